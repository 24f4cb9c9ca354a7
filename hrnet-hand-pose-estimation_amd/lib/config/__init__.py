from .default import _C as cfg
from .default import update_config, get_cfg_defaults
from .node import CfgNode
