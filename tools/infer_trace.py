"""bf16 folded-inference steps for a rocprofv3 kernel trace: python3 tools/infer_trace.py (64 tiles of 256x256x4, 6 forward passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from satellite_computervision_amd import model_tools as mt
mt.reset_uids(); mt.set_seed(0); mt.set_compute_dtype('bfloat16')
m = mt.get_unet_model(2, 4)
x = torch.from_numpy(np.random.default_rng(0).beta(2, 5, (64, 256, 256, 4)).astype(np.float32)).cuda()
if len(sys.argv) > 1 and sys.argv[1] == 'fp8':
    m.enable_fp8_inference(x[:8])
for _ in range(6):
    m.predict_on_device(x)
torch.cuda.synchronize()
