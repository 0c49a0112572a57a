#!/bin/bash
# on the GPU box: bench lines of BASELINE.json's other configurations and the secondary workloads (one JSON line each)
R=${R:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
: > $O/${R}_bench_other_configs.jsonl
run() { echo "# $1" >> $O/${R}_bench_other_configs.jsonl; shift; timeout 600 python bench.py --no-cpu-baseline "$@" 2>> $O/bench.err | tail -1 >> $O/${R}_bench_other_configs.jsonl; }
run "config 2: 53138-triangle mesh @256x256 with anti-aliasing (S = 512), 1 view" --mesh-n 164 --image-size 256 --views-per-gpu 1 --anti-aliasing
run "config 2 without anti-aliasing" --mesh-n 164 --image-size 256 --views-per-gpu 1
run "config 4 per-GPU shard at 4 GPUs: 8 of the 32 cameras" --views-per-gpu 8
run "config 5: 1002528-triangle mesh @1024x1024, 8 views" --mesh-n 709 --image-size 1024 --views-per-gpu 8
run "config 5 per-GPU shard at 8 GPUs: 32 of the 256 cameras" --mesh-n 709 --image-size 1024 --views-per-gpu 32 --steps 10
run "config 4 per-GPU shard at 2 GPUs: 16 of the 32 cameras" --views-per-gpu 16
run "config 4 per-GPU shard at 8 GPUs: 4 of the 32 cameras" --views-per-gpu 4
run "64 views per GPU" --views-per-gpu 64
run "config 3: gan2shape renderer block, batch 16" --workload gan2shape
run "config 3 with flip3: batch 32" --workload gan2shape --flip
run "face3d mesh_cython family" --workload mesh_family
run "silhouettes mode (render_silhouettes + loss + backward), 32 views" --workload silhouettes
run "depth mode (render_depth + masked L1 + backward), 32 views" --workload depth
run "silhouettes mode with anti-aliasing (S = 1024), 32 views" --workload silhouettes --anti-aliasing
run "config 4's mesh @1024x1024, 8 views" --image-size 1024 --views-per-gpu 8
run "the headline with anti-aliasing (Renderer's default): 32 views, output 512x512, internal S = 1024" --anti-aliasing
run "silhouettes mode, config 4's 4-view shard" --workload silhouettes --views-per-gpu 4
run "the unmodified caller's step as the timed line: render() with nothing registered + loss operators (--materialise-images)" --materialise-images --no-dropin
run "... with the losses in eager torch operators" --materialise-images --loss-form torch --no-dropin
# the N > 1 step on ONE rank through RCCL (D3M_BENCH_FORCE_DIST: process group, the split exchange's two all-reduces per step)
rccl() { echo "# $1" >> $O/${R}_bench_other_configs.jsonl; shift; D3M_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --no-cpu-baseline --no-dropin "$@" 2>> $O/bench.err | grep "^{" | tail -1 >> $O/${R}_bench_other_configs.jsonl; }
rccl "one rank through RCCL, split exchange (two graphs, two all-reduces): 32 views"
D3M_SPLIT_EXCHANGE=force rccl "one rank through RCCL, split exchange (forced: the default keeps the one-graph step at this size): config 4's 8-view shard" --views-per-gpu 8
D3M_SPLIT_EXCHANGE=0 rccl "one rank through RCCL, one all-reduce behind the step: 32 views"
D3M_SPLIT_EXCHANGE=0 rccl "one rank through RCCL, one all-reduce: config 4's 8-view shard" --views-per-gpu 8
python3 - <<PY
import json
for l in open("$O/${R}_bench_other_configs.jsonl"):
    if l.startswith("#"): print(l.strip()); continue
    d=json.loads(l); print("   ", d["value"], d["unit"], d["ms_per_step"], "ms")
PY
# config 5 (the bidding form of coverage): kernel stats of the 8-view line
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o stats -- python3 bench.py --no-cpu-baseline --no-dropin --mesh-n 709 --image-size 1024 --views-per-gpu 8 --steps 20 > $O/c5_stats.log 2>&1
cp $(find $O/c5_stats -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats_config5.csv
head -8 $O/${R}_kernel_stats_config5.csv | cut -c1-140
