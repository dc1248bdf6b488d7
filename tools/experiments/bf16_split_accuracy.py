"""How close a product of fp32 operands gets on the bf16 matrix path (DESIGN §8 "what comes next", item 3) — numpy on the host, no GPU.

Every fp32 value a is split into three bf16 pieces a = hi + mid + lo (round-to-nearest-even each time, the remainder carried on);
an fp32 x fp32 product is then the sum of partial bf16 x bf16 products, each exact in fp32, accumulated in fp32 like the MFMA does:
    3 terms   hi.hi + hi.mid + mid.hi                     (what "bf16x3" libraries issue)
    6 terms   + hi.lo + lo.hi + mid.mid                   (everything above 2^-24 relative)
Compared on the conv stack's contraction lengths with activations / weights of the NatureCNN's scale, against an f64 product; the
plain fp32 product (fp32 FMA chain, what v_mfma_f32_16x16x4_f32 computes) is the yardstick.

    python tools/experiments/bf16_split_accuracy.py
"""
import numpy as np


def bf16(x):
    """Round fp32 to bf16 (nearest even), returned as fp32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16(r1)
    lo = bf16((r1 - mid).astype(np.float32))
    return hi, mid, lo


def matmul_f32_chain(a, b, slab=4):
    """fp32 accumulation in k order, `slab` products at a time (the MFMA's k = 4 steps)."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(0, a.shape[1], slab):
        acc = (acc + a[:, k:k + slab].astype(np.float32) @ b[k:k + slab].astype(np.float32)).astype(np.float32)
    return acc


def main():
    rs = np.random.RandomState(0)
    print(f'{"K":>6s} {"fp32 chain":>12s} {"bf16 x3":>12s} {"bf16 x6":>12s}   (max |err| / max |exact| over a 64 x 64 tile)')
    for K in (64, 256, 512, 576, 3136):
        a = np.maximum(rs.standard_normal((64, K)), 0).astype(np.float32)            # post-ReLU activations
        b = (rs.standard_normal((K, 64)) * np.sqrt(2.0 / K)).astype(np.float32)      # weights at init scale
        exact = a.astype(np.float64) @ b.astype(np.float64)
        scale = np.abs(exact).max()
        ah, am, al = split3(a)
        bh, bm, bl = split3(b)
        x3 = matmul_f32_chain(ah, bh, 32) + matmul_f32_chain(ah, bm, 32) + matmul_f32_chain(am, bh, 32)
        x6 = x3 + matmul_f32_chain(ah, bl, 32) + matmul_f32_chain(al, bh, 32) + matmul_f32_chain(am, bm, 32)
        f = matmul_f32_chain(a, b)
        err = lambda c: np.abs(c.astype(np.float64) - exact).max() / scale  # noqa: E731
        print(f'{K:6d} {err(f):12.2e} {err(x3):12.2e} {err(x6):12.2e}')


if __name__ == '__main__':
    main()
