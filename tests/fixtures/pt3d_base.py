work_dir = "results/example_pt3d"
model = dict(type='imgs2mesh', model_cfgs=dict(model_name="imgs2face", image_size=256, texture_size=256, tuplesize=3))
