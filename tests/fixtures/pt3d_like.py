# deep3dmap python-config format, keys as configs/pt3d_demos/imgs2face_multipie.py:28-29 uses them.
_base_ = 'pt3d_base.py'
model = dict(model_cfgs=dict(image_size=64, texture_size=64))
