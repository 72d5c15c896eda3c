"""ggml block-format table (host side).

Type ids: HK/ggml/ggml-common.h:1128-1161 of the reference; block sizes from the
struct definitions at HK/ggml/ggml-common.h:17-108.
"""
from enum import IntEnum


class GGMLType(IntEnum):
    Q4_0 = 2
    Q4_1 = 3
    Q5_0 = 6
    Q5_1 = 7
    Q8_0 = 8
    Q8_1 = 9
    Q2_K = 10
    Q3_K = 11
    Q4_K = 12
    Q5_K = 13
    Q6_K = 14
    # the IQ formats: dequantise + MMVQ only (HK/ggml/ggml_kernel.cu:145-189)
    IQ2_XXS = 16
    IQ2_XS = 17
    IQ3_XXS = 18
    IQ1_S = 19
    IQ4_NL = 20
    IQ3_S = 21
    IQ2_S = 22
    IQ4_XS = 23
    IQ1_M = 29


# type -> (elements per block, bytes per block)
BLOCK = {
    GGMLType.Q4_0: (32, 18), GGMLType.Q4_1: (32, 20), GGMLType.Q5_0: (32, 22),
    GGMLType.Q5_1: (32, 24), GGMLType.Q8_0: (32, 34), GGMLType.Q8_1: (32, 36),
    GGMLType.Q2_K: (256, 84), GGMLType.Q3_K: (256, 110), GGMLType.Q4_K: (256, 144),
    GGMLType.Q5_K: (256, 176), GGMLType.Q6_K: (256, 210),
    GGMLType.IQ4_NL: (32, 18), GGMLType.IQ4_XS: (256, 136),
    GGMLType.IQ2_XXS: (256, 66), GGMLType.IQ2_XS: (256, 74), GGMLType.IQ2_S: (256, 82), GGMLType.IQ3_XXS: (256, 98),
    GGMLType.IQ3_S: (256, 110), GGMLType.IQ1_S: (256, 50), GGMLType.IQ1_M: (256, 56),
}

WEIGHT_TYPES = [GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_0, GGMLType.Q5_1, GGMLType.Q8_0,
                GGMLType.Q2_K, GGMLType.Q3_K, GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q6_K]
# formats with dequantise + MMVQ kernels only (the reference's ggml_mul_mat_a8 has no case for them)
IQ_TYPES = [GGMLType.IQ4_NL, GGMLType.IQ4_XS, GGMLType.IQ2_XXS, GGMLType.IQ2_XS, GGMLType.IQ2_S, GGMLType.IQ3_XXS,
            GGMLType.IQ3_S, GGMLType.IQ1_S, GGMLType.IQ1_M]
# formats whose MMQ activation scratch stores half2(d, sum) — mmq_need_sum, HK/ggml/mmq.cu:84-106
NEED_SUM = {GGMLType.Q4_0, GGMLType.Q4_1, GGMLType.Q5_1, GGMLType.Q4_K, GGMLType.Q5_K}


def block_elems(t):
    return BLOCK[GGMLType(int(t))][0]


def block_bytes(t):
    return BLOCK[GGMLType(int(t))][1]


def row_bytes(t, k):
    qk, bs = BLOCK[GGMLType(int(t))]
    if k % qk:
        raise ValueError(f"k={k} is not a multiple of the {GGMLType(int(t)).name} block size {qk}")
    return k // qk * bs


def weight_bytes(t, n_rows, k):
    return n_rows * row_bytes(t, k)
