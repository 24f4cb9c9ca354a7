"""critical path of the recorded programs given isolated per-op durations (unlimited-GPU model)"""
import sys, os, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import bench as B
from hipnet import _capi as C
model, cfg = B.build_model(torch.bfloat16, 'RHD_HRNet_w32_bf16_train.yaml') if hasattr(B, 'build_model') else None
