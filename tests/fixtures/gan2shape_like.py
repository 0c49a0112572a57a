# A config in deep3dmap's python-config format, written for these tests with the renderer keys the
# gan2shape configs set (configs/gan2shape/celeba.py:30,42-43,69-71).  Not a copy of any reference file.
work_dir = "results/example"
distributed = False
dist_params = dict(backend='nccl')
image_size = 32
model = dict(
    type='Gan2Shape',
    model_cfgs=dict(
        model_name="example", image_size=image_size, batchsize=4,
        min_depth=0.9, max_depth=1.1, rot_center_depth=1.0, fov=10, tex_cube_size=2,
    ))
data = dict(samples_per_gpu=4)
