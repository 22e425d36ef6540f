"""How exact are BatchNorm batch statistics of a 1x1 convolution's output when they are computed from the Gram matrix of its INPUT
(mean_c = w_c . mean(a), E[y_c^2] = w_c' (A'A / M) w_c) instead of from the output itself?  (statistics that do not wait
for conv3's output would let its epilogue write the block output directly and save a 51-MB write + read per bottleneck block.)
CPU only: python3 scripts/probe_gram_bn_stats.py"""
import torch

torch.manual_seed(0)
M, K, N = 12544, 256, 1024
for bias in (0.0, 0.5, 2.0):
    a = torch.relu(torch.randn(M, K) + bias).float()                       # post-ReLU activations (mean/std ratio grows with bias)
    w = (torch.randn(N, K) / K ** 0.5).float()
    y = a @ w.t()                                                          # fp32 outputs, as the convolution stores them
    y64 = a.double() @ w.double().t()
    # reference statistics: what the pipeline uses today (fp32 outputs, fp64 accumulation of per-tile fp32 sums ~ fp64 here)
    mean_ref, ex2_ref = y.double().mean(0), (y.double() ** 2).mean(0)
    var_ref = ex2_ref - mean_ref ** 2
    # Gram route: column sums and A'A accumulated in fp32 inside 64-row blocks (what a matrix-core kernel would do), fp64 across blocks
    blocks = a.view(M // 64, 64, K)
    s = blocks.sum(1).double().sum(0)                                       # [K]
    G = torch.einsum("bik,bil->bkl", blocks, blocks).double().sum(0)        # fp32 products and sums per block, fp64 across
    wd = w.double()
    mean_g = (wd @ s) / M
    ex2_g = torch.einsum("nk,kl,nl->n", wd, G, wd) / M
    var_g = ex2_g - mean_g ** 2
    inv_ref, inv_g = 1 / torch.sqrt(var_ref + 1e-5), 1 / torch.sqrt(var_g + 1e-5)
    # effect on a normalised output value of typical size: y_hat = (y - mean) * invstd
    yh_ref = (y64 - mean_ref) * inv_ref
    yh_g = (y64 - mean_g) * inv_g
    print(f"input bias {bias}: |mean| / std of the outputs up to {float((mean_ref.abs() / var_ref.sqrt()).max()):.2f}; "
          f"rel. diff of invstd max {float(((inv_g - inv_ref) / inv_ref).abs().max()):.2e}; "
          f"max |d normalised output| {float((yh_g - yh_ref).abs().max()):.2e} "
          f"(fp32 rounding of the outputs themselves moves it by {float(((y.double() - y64) * inv_ref).abs().max()):.2e})")
