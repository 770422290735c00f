"""Build recipe for librag4dyg_hip.so (gfx950 only): hipcc, in-tree, no JIT cache.

    python -m rag4dyg_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "_build")
LIB = os.path.join(PKG, "librag4dyg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(PKG), "include", "r4d.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, hdr_m):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_m):
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


# Cross-workgroup hand-offs (topk.hip last-arriver merge, gemm_skinny.hip split-K combine) publish with RELAXED agent-scope
# atomic stores and consume with relaxed agent-scope loads: correct on gfx950 because hipcc lowers those to write-through /
# L1-bypassing `sc1` memory instructions (MI355X_MICROARCH.md, inter-workgroup visibility).  A toolchain that stopped emitting
# sc1 would corrupt top-k results and decode projections silently, so the build fails instead.
# kernel -> (publish stores, consume loads) its source has: topk publishes TWO arrays (values, indices) from two places (winners,
# padding) and consumes both; the skinny GEMM publishes its partial tile in two places and consumes it in one.  EVERY one of them
# must come out as an sc1 instruction -- "some sc1 somewhere in the kernel" let a toolchain pass that lowered only one of the two
# arrays
HANDOFF_KERNELS = {"topk.hip": {"topk_chunk_kernelIfE": (4, 2), "topk_chunk_kernelIdE": (4, 2)},
                   "gemm_skinny.hip": {"gemm_skinny16_kernelILi2ELi0E": (2, 1), "gemm_skinny16_kernelILi3ELi0E": (2, 1)}}


def check_handoff_lowering(src):
    import re
    import tempfile
    names = HANDOFF_KERNELS[os.path.basename(src)]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "--offload-device-only", "-S", src, "-o", asm],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr}")
        text = open(asm).read()
    for name, (want_st, want_ld) in names.items():
        m = re.search(r"^(_ZN3r4d\d+" + re.escape(name) + r"[^\n:]*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
        if not m:
            raise RuntimeError(f"hand-off check: kernel {name} not found in the assembly of {src}")
        body = m.group(2)
        stores = re.findall(r"global_store_dword(?:x2)? [^\n]*", body)
        loads_sc1 = [ln for ln in re.findall(r"global_load_dword(?:x2)? [^\n]*", body) if " sc1" in ln]
        stores_sc1 = [ln for ln in stores if " sc1" in ln]
        ticket = re.findall(r"global_atomic_add[^\n]*", body)
        if not ticket or len(stores_sc1) < want_st or len(loads_sc1) < want_ld:
            raise RuntimeError(f"hand-off check FAILED for {name}: {len(stores_sc1)} sc1 stores (want {want_st}), {len(loads_sc1)} sc1 loads (want {want_ld}), "
                               f"{len(ticket)} ticket atomics -- the relaxed agent-scope publish / consume of {os.path.basename(src)} "
                               "is no longer lowered to write-through / L1-bypassing instructions; use release / acquire there")


# Packed-fp32 instructions with an SGPR operand in the f16x2 kernels (ADVICE r4 medium / VERDICT r4 item 8a).  Round 4 saw a
# compiler-formed v_pk_fma_f32 with an SGPR-pair multiplier give a wrong high half in the lanes (lane & 12) == 12 of an epilogue
# variant that was then removed.  tools/pk_fma_probe.hip (run by tests/test_gpu_dispatch.py on the GPU) checks every operand
# FORM below against scalar arithmetic straight behind MFMAs on gfx950: all agree, bit-reproducibly -- the instruction forms are
# exonerated (the lanes that were wrong are exactly the lanes whose LDS-image swizzle class (li >> 2) & 3 is 3: that variant's own
# fragment addressing, gone with it).  What stays is a census: a shipped kernel may contain only forms the probe has covered
# (register numbers normalised; SGPR operand broadcast from its low word, or a genuine pair).  A new form fails the build until
# the probe covers it.
PACKED_F32_FILES = ("gemm_h2.hip", "attention_h2.hip")
PACKED_F32_FORMS = {
    "v_pk_mul_f32 V, V, S op_sel_hi:[1,0]", "v_pk_mul_f32 V, S, V op_sel_hi:[0,1]", "v_pk_mul_f32 V, V, S", "v_pk_mul_f32 V, S, V",
    "v_pk_fma_f32 V, V, S, V op_sel_hi:[1,0,1]", "v_pk_fma_f32 V, S, V, V op_sel_hi:[0,1,1]", "v_pk_fma_f32 V, V, S, V", "v_pk_fma_f32 V, S, V, V",
    "v_pk_fma_f32 V, V, S, V op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]",
    "v_pk_fma_f32 V, V, S, V op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]",
    "v_pk_fma_f32 V, S, V, V op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]",
    "v_pk_add_f32 V, V, S op_sel_hi:[1,0]", "v_pk_add_f32 V, S, V op_sel_hi:[0,1]",
}


def check_packed_f32_forms(src):
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "--offload-device-only", "-S", src, "-o", asm],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed for {src}:\n{r.stderr}")
        text = open(asm).read()
    seen = {}
    for ln in re.findall(r"^\s*(v_pk_(?:fma|mul|add)_f32 [^\n;]*)", text, re.M):
        if not re.search(r"\bs\[?\d", ln):
            continue
        form = re.sub(r"a\[[0-9:]+\]", "V", re.sub(r"v\[[0-9:]+\]", "V", re.sub(r"\bs\[[0-9:]+\]|\bs\d+\b", "S", ln))).strip()
        seen[form] = seen.get(form, 0) + 1
    new = {f: n for f, n in seen.items() if f not in PACKED_F32_FORMS}
    if new:
        raise RuntimeError(f"packed-fp32 census FAILED for {os.path.basename(src)}: operand forms with an SGPR source that "
                           f"tools/pk_fma_probe.hip has not verified on gfx950: {new} -- add the form to the probe, run it on the GPU, "
                           "then list it in rag4dyg_amd/build.py:PACKED_F32_FORMS (or pin that arithmetic scalar)")
    return seen


def build_library(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr_m), sources()))
    objs = [o for o, _ in res]
    for src, (_o, compiled) in zip(sources(), res):
        if compiled and os.path.basename(src) in HANDOFF_KERNELS:
            check_handoff_lowering(src)
        if compiled and os.path.basename(src) in PACKED_F32_FILES:
            check_packed_f32_forms(src)
    # (also when an earlier run compiled objects but stopped before the link, e.g. on a failed hand-off check)
    stale = os.path.exists(LIB) and any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if force or stale or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB}")
    elif verbose:
        print(f"up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
