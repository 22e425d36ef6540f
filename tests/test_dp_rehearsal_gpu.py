"""GPU: rehearsal of the N>1 path of bench.py / engine on a ONE-GPU box: two ranks share cuda:0 and exchange
gradients over gloo (RCCL refuses two ranks on one device).  Everything except the transport is the code the driver
runs at N=2,4,8: the `python bench.py --gpus N` self-launch, rank != 0 branches, bucketed overlapped exchange with device
tensors, token-weighted gradient scaling, lock-stepped extra steps."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHARE = dict(DIC_DIST_BACKEND="gloo", DIC_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")


def _json_line(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def test_bench_gpus2_self_launch_over_gloo_on_one_gpu(lib):
    """`python bench.py --gpus 2` with NO launcher on the command line must itself start two ranks (the way the driver's
    N=1 command line would reach N>1) and report them."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SHARE)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--no-alt-mode"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["config"]["global_batch"] == 16
    assert d["config"]["batch_per_gpu"] == 8 and d["scaling"] == "weak" and "all-reduce" in d["config"]["collective"]
    assert d["value"] > 0 and d["loss"] == d["loss"]            # finite
    assert d["config"]["collective_ranks"] == 2 and "ok" in d["config"]["collective_self_check"]


@pytest.mark.parametrize("full_size", [False, True])
def test_two_rank_step_equals_one_rank_emulation(lib, tmp_path, full_size, monkeypatch):
    """N-rank == 1-rank: two real ranks (own process each, gloo all-reduce of the two gradient buckets) take one training
    step on the two halves of a RAGGED length-sorted global batch; a single process then replays the same step shard by
    shard (virtual_world=2: same per-shard BatchNorm statistics = DDP semantics, same token-weighted gradient scaling),
    sums the two gradient buffers and applies AdamW.  Post-step parameters must agree to 1e-5 (all-reduce vs in-order sum
    differ in nothing but the transport), both ranks must hold identical parameters.
    full_size: BASELINE config 3 per rank - 2 ranks x 32 images at 224x224, ResNet-152, seq-len 20, V = 10 000 (the 8-GPU
    configuration's per-rank shape; the transport is gloo because two ranks share this box's one GPU)."""
    import importlib
    import socket
    from depth_image_captioning_pub_amd.engine import CaptionTrainer
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    monkeypatch.setenv("DP_FULL_SIZE", "1" if full_size else "0")
    import dp_step_worker as wk
    wk = importlib.reload(wk)

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SHARE)
    env["DP_FULL_SIZE"] = "1" if full_size else "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_step_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ranks = [torch.load(tmp_path / f"rank{i}.pt") for i in range(2)]
    assert torch.equal(ranks[0]["params"], ranks[1]["params"]), "ranks diverged after the all-reduced step"
    assert ranks[0]["drop_seed"] != ranks[1]["drop_seed"], "every rank must draw its own dropout masks"
    if not full_size:
        assert ranks[0]["tokens"] > ranks[1]["tokens"]                  # ragged: rank 0 holds the longer captions

    tr = CaptionTrainer(wk.VOCAB, device="cuda:0", seed=7, resnet_layers=wk.LAYERS, conv_mode="bf16x3")
    gsum, losses = None, []
    for rank in range(2):
        imgs, depth, caps, ln, drop, gtok = wk.shard(rank, 2)
        loss = tr.train_step(imgs.cuda(), depth.cuda(), caps.cuda(), ln, drop_mult=drop.cuda(), global_tokens=gtok,
                             virtual_world=2, apply_update=False)
        g = tr.flat.grad.clone()
        gsum = g if gsum is None else gsum + g
        losses.append(float(loss.item()))
    tr.flat.grad.copy_(gsum)
    tr.apply_update()
    torch.cuda.synchronize()
    for rank in range(2):
        assert abs(ranks[rank]["loss"] - losses[rank]) <= 1e-5, (rank, ranks[rank]["loss"], losses[rank])
    err = float((tr.flat.data.cpu() - ranks[0]["params"]).abs().max())
    assert err <= 1e-5, f"2-rank parameters differ from the 1-rank emulation by {err:.3e}"


def test_bench_hard_workload_line(lib):
    """`bench.py --hard` (the depth-hard step of BASELINE config 4 as a separate workload line, never the headline): runs, names
    itself, streams all 196 cells, and its same-run parity gate (CPU oracle's depth-hard step on the same Gumbel draws) is green."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--hard", "--steps", "3", "--warmup", "2", "--batch", "16", "--no-alt-mode",
           "--cpu-batch", "8", "--cpu-iters", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert "depth-hard" in d["metric"] and "Gumbel" in d["config"]["workload"] and d["config"]["annotation_cells"] == 196
    assert d["value"] > 0 and d["parity"]["ok"] and d["parity"]["annotation_cells"] == 196
    assert "depth-hard" in d["cpu_baseline"]["sample"]


def test_engine_exchange_over_rccl_single_rank(lib):
    """engine.exchange_gradients over a real RCCL process group (`torch.distributed` backend "nccl", one rank on cuda:0): the
    collective code path of the data-parallel step executes on RCCL here - with one rank, which is all a one-GPU box admits;
    the N-rank behaviour is covered over gloo (two ranks sharing this GPU, and world 2 / 4 on the CPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_rccl_worker.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_EXCHANGE_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_rccl_entry_points_of_the_c_abi_single_rank(lib):
    """dic_comm_unique_id / dic_comm_create / dic_allreduce_grads / dic_comm_destroy (include/dic.h, "data parallel"): the
    gradient exchange for callers that bind the C ABI without torch.distributed.  A one-GPU box admits one RCCL rank per
    device, so this checks what can be checked here: RCCL resolves at run time (the copy this process already holds, i.e.
    PyTorch's), a 1-rank communicator comes up on cuda:0, a sum all-reduce of a gradient-sized flat buffer (6.5 M floats) on
    the caller's stream returns the buffer bit for bit, errors come back as codes with a message."""
    import ctypes as C
    from depth_image_captioning_pub_amd._lib import check, ptr, stream_ptr
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")
    ident = (C.c_char * 128)()
    check(lib.dic_comm_unique_id(ident), "dic_comm_unique_id")
    assert any(b != 0 for b in bytes(ident))
    comm = C.c_void_p()
    check(lib.dic_comm_create(ident, 1, 0, C.byref(comm)), "dic_comm_create")
    try:
        n, r = C.c_int(-1), C.c_int(-1)
        check(lib.dic_comm_ranks(comm, C.byref(n), C.byref(r)), "dic_comm_ranks")
        assert (n.value, r.value) == (1, 0)
        g = torch.randn(6_471_104, device="cuda:0")
        ref = g.clone()
        check(lib.dic_allreduce_grads(comm, ptr(g), C.c_longlong(g.numel()), stream_ptr()), "dic_allreduce_grads")
        torch.cuda.synchronize()
        assert torch.equal(g, ref)
        assert lib.dic_allreduce_grads(comm, None, C.c_longlong(4), stream_ptr()) != 0
        assert b"bad arguments" in lib.dic_last_error()
    finally:
        check(lib.dic_comm_destroy(comm), "dic_comm_destroy")
    assert lib.dic_comm_create(ident, 2, 5, C.byref(comm)) != 0          # rank outside [0, nranks)
