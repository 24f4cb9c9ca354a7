from . import pose_hrnet  # noqa: F401  (the reference dispatches eval(cfg.MODEL.NAME + '.get_pose_net'))
