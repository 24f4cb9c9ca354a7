from . import pose_hrnet, pose_hrnet_PoseAggr, pose_hrnet_softmax  # noqa: F401  (the reference dispatches eval(cfg.MODEL.NAME + '.get_pose_net'))
