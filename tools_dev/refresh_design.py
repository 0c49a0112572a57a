"""Rewrites the measured table and the figures of DESIGN.md 4.5 / 7 from the committed profiles/r02_* artefacts."""
import csv, importlib.util, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
s = open('DESIGN.md').read()
st = list(csv.DictReader(open('profiles/r02_kernel_stats_final.csv')))
avg = lambda name: next(float(r['AverageNs']) / 1e3 for r in st if name in r['Name'])
pm = json.load(open('profiles/r02_pmc_traffic_final.json'))['kernels']
pmget = lambda k: next(pm[n]['hbm_bytes_per_launch'] for n in pm if n.startswith(k))
L = lambda f: json.loads(open('profiles/' + f).read().strip().splitlines()[-1])
fin, ser, mat = L('r02_bench_final.json'), L('r02_bench_serial_branches.json'), L('r02_bench_materialise_images.json')
sk = ser['kernel_ms_per_step']
spec = importlib.util.spec_from_file_location('bench', 'bench.py'); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
V, F, S, ts = 50625, 100352, 512, 2
rows = [('k_edge_lines', '`k_edge_lines` (K4: set-up + walks)'), ('k_raster_tiles', '`k_raster_tiles`'),
        ('k_backward_textures_lit_faces', '`k_backward_textures_lit_faces` (K5+K6, side branch)'),
        ('k_render_lit_fit_records', '`k_render_lit_fit_records` (sampling + objective + walk records)'),
        ('k_edge_scatter', '`k_edge_scatter` (plan records, side branch)'), ('k_edge_count', '`k_edge_count` (side branch)'),
        ('k_edge_gather', '`k_edge_gather`'), ('k_bin_count', '`k_bin_count` (incl. the face gather)'), ('k_bin_fill', '`k_bin_fill`')]
tab = ('| kernel | µs in the step (rocprofv3 avg, branches concurrent) | µs on its own (`D3M_SERIAL_BRANCHES`) | algorithmic MB / launch '
       '| PMC HBM MB / launch (2·FETCH+WRITE) |\n|---|---|---|---|---|\n')
for k, label in rows:
    tab += f"| {label} | {avg(k):.0f} | {sk.get(k, 0) * 1000:.0f} | {bench.kernel_bytes(k, V, F, S, ts) * 32 / 1e6:.0f} | {pmget(k) / 1e6:.0f} |\n"
oc = sum(float(r['TotalDurationNs']) for r in st if not any(k in r['Name'] for k, _ in rows) and 'k_render_lit_epilogue' not in r['Name']) / 49 / 1e3
os_ = sum(v for k, v in sk.items() if k not in dict(rows)) * 1000
tab += f"| everything else (26 launches: visibility list, scans, cameras, light, fills, view sum) | ≈ {oc:.0f} | ≈ {os_:.0f} | | |\n"
a, b = s.index('| kernel | µs in the step (rocprofv3 avg, branches concurrent)'), s.index('**Step: ')
s = s[:a] + tab + '\n' + s[b:]
s = re.sub(r'\*\*Step: [0-9.]+ ms = [0-9]+ Mpix/s\*\* in the committed line', f"**Step: {fin['ms_per_step']:.2f} ms = {fin['value']:.0f} Mpix/s** in the committed line", s)
s = re.sub(r'the same step takes [0-9.]+ ms: \*\*the side branches buy', f"the same step takes {ser['ms_per_step']:.2f} ms: **the side branches buy", s)
r, rs = fin['roofline'], ser['roofline']
s = re.sub(r'[0-9]+ MB / [0-9]+ µs = [0-9]+ GB/s = [0-9.]+ % of the 8 TB/s HBM peak \([0-9.]+ % on its own\), with',
           f"{r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB / {r['avg_launch_us']:.0f} µs = {r['achieved']:.0f} GB/s = {r['frac'] * 100:.1f} % of the 8 TB/s HBM peak ({rs['frac'] * 100:.1f} % on its own), with", s)
s = re.sub(r'`valu_issue_frac` [0-9.]+ from the SQ counters \([0-9.]+ on its own\)', f"`valu_issue_frac` {r['valu_issue_frac']} from the SQ counters ({rs['valu_issue_frac']} on its own)", s)
s = re.sub(r'Whole step: 1.56 GB algorithmic / [0-9.]+ ms = [0-9.]+ % of HBM peak', f"Whole step: 1.56 GB algorithmic / {fin['ms_per_step']:.2f} ms = {fin['hbm_roofline_frac_step'] * 100:.1f} % of HBM peak", s)
s = re.sub(r'the final build takes [0-9.]+ ms = [0-9]+ Mpix/s \(`profiles/r02_bench_materialise_images.json`',
           f"the final build takes {mat['ms_per_step']:.2f} ms = {mat['value']:.0f} Mpix/s (`profiles/r02_bench_materialise_images.json`", s)
open('DESIGN.md', 'w').write(s)
print(fin['value'], fin['ms_per_step'], ser['ms_per_step'], mat['ms_per_step'])
