"""Developer aid: libquilt_amd first, torch second (the load order that used to leave torch without a device)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quilt_amd import native
print("devices seen by libquilt_amd:", native.lib().qa_device_count())
import torch
print("torch.cuda.is_available():", torch.cuda.is_available())
x = torch.ones(4, device="cuda") * 2
print("torch op:", x.sum().item())
from quilt_amd.synth import make_synthetic_panel
from quilt_amd.native import DevicePanel
dev = DevicePanel(make_synthetic_panel(K=500, nSNPs=320, seed=1))
print("panel on device ok")
