"""rocprofv3 target: ResNet-152 forward (batch 64, train BN) x10 under one debug code.  usage: prof_resnet_code.py CODE"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib
lib = _lib.load()
rn = {k: v.to("cuda:0") for k, v in syn.resnet152_weights(seed=125).items()}
runner = native.ResNetRunner(rn, conv_mode="bf16x3")
imgs = syn.rgb_images(64, seed=123).to("cuda:0")
out = torch.empty((64, 196, 2048), device="cuda:0")
for c in sys.argv[1:]: lib.dic_debug_force_staged_gemm(int(c))
for _ in range(10): runner.forward(imgs, train_bn=True, out=out)
torch.cuda.synchronize()
