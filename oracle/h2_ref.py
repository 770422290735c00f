"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): a numpy restatement of the two-term fp16 form the f16x2 kernels compute in
(``rag4dyg_amd/csrc/h2.h``, ``gemm_h2.hip``, ``attention_h2.hip``) -- not a reference function (the reference computes in fp32:
``models/modeling_utils.py:1267-1271``, ``models/modeling_gpt2.py:140-160``), but the statement of WHY those kernels reproduce
it: every fp32 operand is ``hi + 2^-11 lo'`` to 2^-22, fp16 products are exact in an fp32 accumulator, and three (plane form) or
two-per-8-elements (word form) matrix instructions evaluate ``hi.hi + 2^-11 (hi.lo' + lo'.hi)``.
"""
import numpy as np

LO_SCALE = 2048.0                       # 2^11


def split(x, prescale=1.0):
    """fp32 array -> (hi, lo') fp16 arrays of x * prescale: hi = RN16(x), lo' = RN16((x - hi) 2^11), round to nearest even."""
    v = np.asarray(x, np.float32) * np.float32(prescale)
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(np.float32)) * np.float32(LO_SCALE)).astype(np.float16)
    return hi, lo


def words(x, prescale=0.25):
    """The "h2 word" image of x (uint32): hi in the low half, lo' in the high half, of x * prescale (h2_words<true>)."""
    hi, lo = split(x, prescale)
    return hi.view(np.uint16).astype(np.uint32) | (lo.view(np.uint16).astype(np.uint32) << 16)


def unpack(w):
    """uint32 words -> (hi, lo') as float64."""
    w = np.asarray(w, np.uint32)
    return ((w & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64),
            (w >> 16).astype(np.uint16).view(np.float16).astype(np.float64))


def dot_word_forms(wa, wb, block=8):
    """sum_k a_k b_k as attention_h2.hip evaluates it: operand A = words (k-slots hi, lo'), operand B in the forms
    F1 = (hi, 0) and F2 = (lo', hi); ``block`` elements per instruction, exact products, fp32 accumulation per instruction
    (the hardware adds the 16 slot products of one instruction exactly before rounding into the accumulator -- modelled so)."""
    ha, la = unpack(wa)
    hb, lb = unpack(wb)
    acc0 = np.float32(0.0)
    acc1 = np.float32(0.0)
    for s in range(0, len(ha), block):
        e = slice(s, s + block)
        acc0 = np.float32(acc0 + np.float32(np.sum(ha[e] * hb[e] + la[e] * 0.0)))              # word . F1
        acc1 = np.float32(acc1 + np.float32(np.sum(ha[e] * lb[e] + la[e] * hb[e])))            # word . F2
    return np.float32(acc0 + np.float32(acc1 / LO_SCALE))


def dot_planes(a, b, a_prescale=0.25, block=16):
    """sum_k a_k b_k as gemm_h2.hip evaluates it: three products per 16 k, two fp32 accumulator sets, joined once."""
    ha, la = (t.astype(np.float64) for t in split(a, a_prescale))
    hb, lb = (t.astype(np.float64) for t in split(b))
    acc0 = np.float32(0.0)
    acc1 = np.float32(0.0)
    for s in range(0, len(ha), block):
        e = slice(s, s + block)
        acc1 = np.float32(acc1 + np.float32(np.sum(la[e] * hb[e])))
        acc0 = np.float32(acc0 + np.float32(np.sum(ha[e] * hb[e])))
        acc1 = np.float32(acc1 + np.float32(np.sum(ha[e] * lb[e])))
    return np.float32(np.float32(acc0 + np.float32(acc1 / LO_SCALE)) / np.float32(a_prescale))
