"""Config defaults for the HRNet hot path (key names and semantics follow the
reference's YACS tree, lib/config/default.py:17-257, so its experiments/RHD
yamls merge unchanged; `update_config` mirrors lib/config/default.py:260-270).

Only key NAMES are shared with the reference; keys of model families that are
out of scope keep their defaults so that foreign yamls still merge.
"""
from .node import CfgNode as CN

_DEFAULTS = dict(
    EXP_NAME='', OUTPUT_DIR='', LOG_DIR='', DATA_DIR='', DISTRIBUTED=False, GPUS=(0,), WORKERS=4,
    PRINT_FREQ=20, AUTO_RESUME=False, PIN_MEMORY=True, RANK=0, VERBOSE=True, DIST_BACKEND='nccl',
    MULTIPROCESSING_DISTRIBUTED=False, WITHOUT_EVAL=False, WITH_DATA_AUG=False,
    FP16=dict(ENABLED=False, STATIC_LOSS_SCALE=1.0, DYNAMIC_LOSS_SCALE=True),
    CUDNN=dict(BENCHMARK=True, DETERMINISTIC=False, ENABLED=True),
    MODEL=dict(
        NAME='pose_hrnet', INIT_WEIGHTS=True, PRETRAINED='', TEMPORAL_PRETRAINED='', HRNET_PRETRAINED='',
        NUM_JOINTS=21, TAG_PER_JOINT=True, TARGET_TYPE='gaussian', IMAGE_SIZE=[256, 256],
        HEATMAP_SIZE=[64, 64], SIGMA=2, SYNC_BN=False, HEATMAP_SOFTMAX=False, TRAINABLE_SOFTMAX=False,
        # MI355X build: arithmetic type of the device path ('fp32' | 'bf16'); not a reference key.
        COMPUTE_DTYPE='fp32',
        # keys of out-of-scope model families (kept so their yamls still merge)
        N_HIDDEN=[64, 64, 64, 64], STRIDE=1, FILTER_SIZE=5, LAYER_NORM=1, EMBEDDING_SIZE=512,
        TCN_CHANNELS=1024, FILTER_WIDTHS=[3, 3, 3, 3], TRIANGULATION_MODEL_NAME='alg',
        BACKBONE_NAME='pose_hrnet_volumetric', BACKBONE_MODEL_PATH='', CUBOID_SIZE=500.0, VOLUME_SIZE=64,
        SCALE_KEYPOINTS_3D=0.1, VOLUME_MULTIPLIER=1.0, VOLUME_SOFTMAX=True,
        VOLUME_AGGREGATION_METHOD='softmax', USE_GT_MIDDLEROOT=True, ALG_CONFIDENCES=False,
        VOL_CONFIDENCES=True, DIRECT_OPTIMIZATION=False, N_CRITIC=3, CLIP_VALUE=0.01, AGGRE=True,
        DILATION_RATES=[3, 6, 12, 18, 24], USE_WARPING_TRAIN=True, USE_WARPING_TEST=True, PATCH_SIZE=4,
        EMB_DIM=[96], DROP_RATE=0., DROP_PATH_RATE=0., DEPTHS=[2, 2, 6, 2], NUM_HEADS=[3, 6, 12, 24],
        ABSOLUTE_POSITION_ENCODING=False, FF_TYPE='mlp', VERSION='V2+', HAM_TYPE='NMF', S=1, R=64,
        DUAL_HAM=False, SPATIAL=True, CHEESE_FACTOR=1, ZERO_HAM=True, TRAIN_STEPS=6, EVAL_STEPS=7,
        INV_T=100, ETA=0.9, RAND_INIT=True, BETA=0.1, USE_MASK=False, MAKSED_BLOCKS=0,
    ),
    LOSS=dict(
        USE_OHKM=False, TOPK=8, USE_TARGET_WEIGHT=True, USE_DIFFERENT_JOINTS_WEIGHT=False,
        WITH_HEATMAP_LOSS=True, HEATMAP_LOSS_FACTOR=1.0, WITH_POSE2D_LOSS=False, POSE2D_LOSS_FACTOR=1.0,
        WITH_POSE3D_LOSS=True, POSE3D_LOSS_FACTOR=1.0, WITH_TIME_CONSISTENCY_LOSS=False,
        TIME_CONSISTENCY_LOSS_FACTOR=1.0, WITH_BONE_LOSS=False, BONE_LOSS_FACTOR=1.0,
        WITH_JOINTANGLE_LOSS=False, JOINTANGLE_LOSS_FACTOR=1.0, WITH_VOLUMETRIC_CE_LOSS=False,
        VOLUMETRIC_LOSS_FACTOR=0.01, WITH_KCS_LOSS=False, KCS_LOSS_FACTOR=0.01, WITH_KCS_TC_LOSS=False,
        KCS_TC_LOSS_FACTOR=0.01,
    ),
    DATASET=dict(
        ROOT='', BACKGROUND_DIR='', DATASET=[], TEST_DATASET=[], TRAIN_SET='training', TEST_SET='evaluation',
        DATA_FORMAT='jpg', HYBRID_JOINTS_TYPE='', SELECT_DATA=False, NUM_VIEWS=4, SEQ_IDX=[-2, -1, 0, 1, 2],
        STRIDE=2, NUM_JOINTS=21, INPUT_SIZE=256, OUTPUT_SIZE=[64], MAX_ROTATION=30, MIN_SCALE=0.75,
        MAX_SCALE=1.25, SCALE_TYPE='short', MAX_TRANSLATE=40, FLIP=False, SCALE_FACTOR=0.25, ROT_FACTOR=30,
        PROB_HALF_BODY=0.0, NUM_JOINTS_HALF_BODY=8, COLOR_RGB=False, SIGMA=2, SCALE_AWARE_SIGMA=False,
        BASE_SIZE=256.0, BASE_SIGMA=2.0, INT_SIGMA=False, N_FRAMES=1, FRAME_STRIDE=1, SAMPLE_STRIDE=10,
    ),
    TRAIN=dict(
        LR_FACTOR=0.1, LR_STEP=[3, 6], LR=0.001, LR_SCHEDULE='multi_step', WARMUP_EPOCHS=10,
        PROCESS_FEATURE_LR=0.001, VOLUME_NET_LR=0.001, OPTIMIZER='adam', BN_MOMENTUM=3e-4, MOMENTUM=0.9,
        WD=0.0001, NESTEROV=False, GAMMA1=0.99, GAMMA2=0.0, BEGIN_EPOCH=0, END_EPOCH=140, RESUME=False,
        CHECKPOINT='', IMAGES_PER_GPU=32, SHUFFLE=True,
    ),
    TEST=dict(
        IMAGES_PER_GPU=32, FLIP_TEST=False, POST_PROCESS=False, SHIFT_HEATMAP=False, USE_GT_BBOX=False,
        IMAGE_THRE=0.1, NMS_THRE=0.6, SOFT_NMS=False, OKS_THRE=0.5, IN_VIS_THRE=0.0, COCO_BBOX_FILE='',
        BBOX_THRE=1.0, MODEL_FILE='',
    ),
    DEBUG=dict(DEBUG=False, SAVE_BATCH_IMAGES_GT=False, SAVE_BATCH_IMAGES_PRED=False,
               SAVE_HEATMAPS_GT=False, SAVE_HEATMAPS_PRED=False),
)


def _build():
    root = CN(_DEFAULTS)
    root.MODEL['EXTRA'] = CN(new_allowed=True)
    return root


_C = _build()


def get_cfg_defaults():
    return _C.clone()


def update_config(cfg, args):
    """Merge `args.cfg` (yaml) then `args.opts` (KEY value pairs) and freeze."""
    cfg.defrost()
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(getattr(args, 'opts', None))
    cfg.freeze()
