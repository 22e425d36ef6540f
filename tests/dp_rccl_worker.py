"""Child process of tests/test_dp_rehearsal_gpu.py::test_engine_exchange_over_rccl_single_rank: the engine's bucketed,
overlapped gradient exchange (engine.exchange_gradients, the call train_step makes when world > 1) over a REAL RCCL process
group - `torch.distributed` backend "nccl" - of one rank on cuda:0 (a one-GPU box admits one RCCL rank per device).  Prints OK."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from depth_image_captioning_pub_amd import native, synthetic as syn  # noqa: E402
from depth_image_captioning_pub_amd.engine import CaptionTrainer, exchange_gradients  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        V, B, T = 300, 4, 6
        tr = CaptionTrainer(V, device="cuda:0", seed=5, resnet_layers=(1, 1, 1, 1), conv_mode="bf16x3", process_group=dist.group.WORLD)
        imgs = syn.rgb_images(B, seed=1, size=96).cuda(); depth = syn.depth_maps(B, seed=2, size=96).cuda()
        caps, lens = syn.captions_fixed(B, V, T, seed=3); caps = caps.cuda()
        tr.train_step(imgs, depth, caps, lens, apply_update=False)           # world = 1: no exchange inside; fills flat.grad
        torch.cuda.synchronize()
        before = tr.flat.grad.clone()
        ran = []
        # the exchange exactly as train_step issues it at world > 1: decoder bucket first, a stand-in for the depth-encoder backward
        # (which fills the second bucket) in between, then the second bucket - all on RCCL
        exchange_gradients(tr.flat.grad, [tr.dec_span, tr.enc_span], tr.pg, between=lambda: ran.append(1))
        torch.cuda.synchronize()
        assert ran == [1]
        assert torch.equal(tr.flat.grad, before), "a sum all-reduce over one rank must return the buffer unchanged"
        assert dist.get_backend() == "nccl"
        print("RCCL_EXCHANGE_OK", tr.flat.total)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
