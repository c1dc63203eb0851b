"""Test infrastructure (CPU oracle): numpy restatement of the operand split the conv kernels use for fp32-accurate products on the bf16
matrix cores (virconv_amd/csrc/conv_kernels.hip, split3 / operand type X6).  Only tests import this.

    x = h + m + l   exactly, h / m / l bf16 values (8 significant bits), each obtained by ROUNDING TO NEAREST EVEN (v_cvt_pk_bf16_f32):
        h = bf16(x);  r1 = x - h (exact, |r1| <= 2^-9 |x|);  m = bf16(r1);  l = r1 - m (exact, <= 8 significant bits, |l| <= 2^-18 |x|)
    x * y  ~  l_x h_y + h_x l_y + m_x m_y + m_x h_y + h_x m_y + h_x h_y        (the six terms the kernels issue, small ones first)
    dropped: m_x l_y, l_x m_y (<= 2^-26 |x y| each), l_x l_y (<= 2^-36 |x y|): together below HALF an fp32 ulp of the product.
    (A truncating split -- the first version -- leaves |m| <= 2^-7 |x|, |l| <= 2^-14 |x| and drops up to 2^-20 |x y|.)
"""
import numpy as np


def rne_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bf16 (ties to even), returned as the float32 value with the low 16 bits clear."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    lsb = (u >> np.uint64(16)) & np.uint64(1)
    r = ((u + np.uint64(0x7FFF) + lsb) & np.uint64(0xFFFF0000)).astype(np.uint32)
    return r.view(np.float32)


def trunc_bf16(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def split3(x: np.ndarray, rounding=rne_bf16):
    x = np.asarray(x, dtype=np.float32)
    h = rounding(x)
    r1 = (x - h).astype(np.float32)          # exact in fp32
    m = rounding(r1)
    l = (r1 - m).astype(np.float32)          # exact; representable in bf16
    return h, m, l


def product6(x: np.ndarray, y: np.ndarray, rounding=rne_bf16) -> np.ndarray:
    """The six-term product, every term exact (float64 holds a bf16 x bf16 product exactly), summed in float64."""
    hx, mx, lx = (a.astype(np.float64) for a in split3(x, rounding))
    hy, my, ly = (a.astype(np.float64) for a in split3(y, rounding))
    return lx * hy + hx * ly + mx * my + mx * hy + hx * my + hx * hy
