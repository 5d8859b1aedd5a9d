"""Kernel trace target: DeepLab-v3 (config 3) inference, batch 16 of 512x512x4."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satellite_computervision_amd import model_tools as mt
mt.reset_uids(); mt.set_seed(0); mt.set_compute_dtype('bfloat16')
m = mt.get_deeplabv3_model(2, 4)
x = torch.rand(int(os.environ.get('B', '16')), 512, 512, 4, device='cuda')
for _ in range(int(os.environ.get('PASSES', '4'))):
    m.predict_on_device(x)
torch.cuda.synchronize()
