"""A/B of the 3x3 stride-1 convolution kernels on ONE box: conv_igemm (salun_conv2d_forward / _backward_data) against
the LDS-DMA ring kernel (salun_conv3x3_packed) per layer, forward and backward-data, with the tile variants of the ring
kernel pinned one at a time.  Also checks the two kernels' results against each other (bit-identical by design).
ResNet-18 shapes at batch 256, DDPM shapes at batch 128 with --ddpm."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unlearn_saliency_amd import ops

RESNET = [("l1 64->64 @32", 256, 64, 32, 64), ("l2 128->128 @16", 256, 128, 16, 128),
          ("l3 256->256 @8", 256, 256, 8, 256), ("l4 512->512 @4", 256, 512, 4, 512)]
DDPM = [("128->128 @32", 128, 128, 32, 128), ("256->256 @16", 128, 256, 16, 256), ("512->256 @16", 128, 512, 16, 256),
        ("256->256 @8", 128, 256, 8, 256), ("256->256 @4", 128, 256, 4, 256)]


def timeit(fn, iters=100, warm=150):
    # 150 warm-up launches (> 20 ms): the chip takes 15 - 20 ms of continuous work to reach its sustained clock (blocks of
    # 100 launches of one kernel: 167, 150, 145, 145, ... us — tools/sustained_bench.py); a 3-launch warm-up made the
    # first variant timed after an idle spell look 9 - 13 % slower than the same kernel timed later (round 6)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ddpm", action="store_true")
    ap.add_argument("--cfgs", default="0,1,2,3,4,5")
    ap.add_argument("--wgs", default="2")
    a = ap.parse_args()
    cfgs = [int(c) for c in a.cfgs.split(",")]
    wgs = [int(c) for c in a.wgs.split(",")]
    for name, N, C, H, K in (DDPM if a.ddpm else RESNET):
        x = torch.randn(N, C, H, H, device="cuda")
        w = torch.randn(K, C, 3, 3, device="cuda") * 0.05
        dy = torch.randn(N, K, H, H, device="cuda")
        gf = 2.0 * N * K * H * H * C * 9 / 1e9
        y0 = ops.conv2d_forward(x, w, None, 1, 1, H, H)
        d0 = ops.conv2d_backward_data(dy, w, x.shape, 1, 1)
        t_f = timeit(lambda: ops.conv2d_forward(x, w, None, 1, 1, H, H))
        t_d = timeit(lambda: ops.conv2d_backward_data(dy, w, x.shape, 1, 1))
        t_p = timeit(lambda: (ops.conv3x3_pack(w, False), ops.conv3x3_pack(w, True)))
        print(f"{name:18s} {gf:5.1f} GF | igemm fwd {t_f:6.1f} us {gf / t_f * 1e3:6.1f} TF | dgrad {t_d:6.1f} us {gf / t_d * 1e3:6.1f} TF"
              f" | pack fwd+dgrad images {t_p:5.1f} us", flush=True)
        imf, imd = ops.conv3x3_pack(w, False), ops.conv3x3_pack(w, True)
        for cfg in cfgs:
            for wg in wgs:
                code = cfg | (wg << 8)
                y1 = ops.conv3x3_packed(x, imf, K, cfg=code)
                if y1 is None:
                    print(f"    ring cfg {cfg} wgs/CU {wg}: outside the domain")
                    continue
                d1 = ops.conv3x3_packed(dy, imd, C, cfg=code)
                torch.cuda.synchronize()
                ef = (y1 - y0).abs().max().item(); ed = (d1 - d0).abs().max().item()
                bit = bool(torch.equal(y1, y0) and torch.equal(d1, d0))
                r_f = timeit(lambda: ops.conv3x3_packed(x, imf, K, cfg=code))
                r_d = timeit(lambda: ops.conv3x3_packed(dy, imd, C, cfg=code))
                print(f"    ring cfg {cfg} wgs/CU {wg}: fwd {r_f:6.1f} us {gf / r_f * 1e3:6.1f} TF | dgrad {r_d:6.1f} us {gf / r_d * 1e3:6.1f} TF"
                      f" | max|diff| {ef:.2e} {ed:.2e} bit-identical {bit}", flush=True)
        # epilogue terms
        bias = torch.randn(K, device="cuda"); nb = torch.randn(N, K, device="cuda"); add = torch.randn(N, K, H, H, device="cuda")
        yr = ops.conv2d_forward(x, w, bias, 1, 1, H, H, nbias=nb, addend=add)
        ye = ops.conv3x3_packed(x, imf, K, bias=bias, nbias=nb, addend=add)
        if yr is not None and ye is not None:
            print(f"    epilogue terms: max|diff| {(yr - ye).abs().max().item():.2e} bit-identical {bool(torch.equal(yr, ye))}")
        dd = ops.conv2d_backward_data(dy, w, x.shape, 1, 1, addend=x)
        de = ops.conv3x3_packed(dy, imd, C, addend=x)
        if dd is not None and de is not None:
            print(f"    dgrad + addend: max|diff| {(dd - de).abs().max().item():.2e} bit-identical {bool(torch.equal(dd, de))}")


if __name__ == "__main__":
    main()
