"""Builds and loads libmulut_hip.so (the C ABI of include/mulut.h) with ctypes."""
import ctypes
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_LIBDIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(_LIBDIR, "libmulut_hip.so")
SOURCES = ["mulut_kernels.hip", "mulut_k1.hip", "mulut_detail.hip", "mulut_capi.hip", "mulut_ft.hip", "mulut_eval.hip"]
HEADERS = ["mulut_core.h", "mulut_kernels.h", "mulut_dev.h", "mulut_tube2_asm.inc", os.path.join("..", "..", "include", "mulut.h")]
# -Wno-inline-asm: stage_tube2_kernel names registers ABOVE the register allocator's budget in its asm clobber lists on purpose
# (tools/gen_tube2_asm.py); -Wno-pass-failed: its occupancy attribute is that budget, not an occupancy the kernel reaches
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-inline-asm", "-Wno-pass-failed"]

# every symbol include/mulut.h declares
EXPORTS = [
    "mulut_version", "mulut_strerror", "mulut_last_hip_error", "mulut_create", "mulut_destroy",
    "mulut_configure", "mulut_set_lut", "mulut_pass", "mulut_stage", "mulut_pipeline",
    "mulut_pipeline_rows", "mulut_halo", "mulut_reserve", "mulut_set_stage_timing", "mulut_last_stage_ms", "mulut_last_kernel_ms",
    "mulut_set_tuning", "mulut_kernel_name", "mulut_ft_stage_forward", "mulut_ft_stage_backward", "mulut_ft_quantize", "mulut_ft_quantize_backward",
    "mulut_ft_stage_forward_mask", "mulut_ft_stage_backward_mask", "mulut_eval_ws_doubles", "mulut_eval_y", "mulut_last_detail_counters", "mulut_debug_read",
]

_libs = {}


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def source_hash():
    """sha256 over the kernel sources and headers: ties profiler-derived numbers (profiles/*.json) to the build they
    were measured on -- bench.py reports them only while this hash still matches."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + [x for x in HEADERS if not x.endswith("mulut.h")]):      # device code only: the ABI header declares, it does not compute
        with open(os.path.join(_CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(_CSRC, f)) > t for f in SOURCES + HEADERS)


def audit_tube2_isa(asm_path=None):
    """stage_tube2_kernel keeps the rows of the pass in flight in v88..v127 ACROSS asm statements (tools/gen_tube2_asm.py); the only
    things that keep the compiler out of them are its register budget (amdgpu_waves_per_eu) and the clobber lists, and neither binds a
    short-lived temporary by contract.  So the library is not accepted on trust: this compiles mulut_kernels.hip to gfx950 assembly
    (or reads `asm_path`) and raises unless, in every stage_tube2_kernel instance, (i) no instruction outside the inline-asm blocks
    names v<TUBE2_ROW0> or above, (ii) no AGPR is used and (iii) at most 128 VGPRs are allocated (the kernel is launched with 1024
    threads at 4 waves per SIMD: 512 / 4 registers per lane).  Returns the number of kernels audited."""
    import re
    import tempfile
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot audit stage_tube2_kernel")
    inc = open(os.path.join(_CSRC, "mulut_tube2_asm.inc")).read()
    row0 = int(re.search(r"#define TUBE2_ROW0 (\d+)", inc).group(1))
    with tempfile.TemporaryDirectory() as td:
        if asm_path is None:
            asm_path = os.path.join(td, "mulut_kernels.s")
            flags = [f for f in HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
            subprocess.check_call([hipcc] + flags + ["--cuda-device-only", "-S", "-o", asm_path, os.path.join(_CSRC, "mulut_kernels.hip")],
                                  stderr=subprocess.DEVNULL)
        lines = open(asm_path).read().splitlines()
    reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
    kernels, name, in_asm, bad = [], None, False, []
    for line in lines:
        m = re.match(r"^(_ZN5mulut18stage_tube2_kernel\w+):", line)
        if m:
            name, in_asm = m.group(1), False
            kernels.append(name)
            continue
        m = re.match(r"^\s*\.set (_ZN5mulut18stage_tube2_kernel\w+)\.(num_vgpr|num_agpr), (\d+)", line)
        if m:
            if (m.group(2) == "num_agpr" and int(m.group(3)) != 0) or (m.group(2) == "num_vgpr" and int(m.group(3)) > 128):
                bad.append((m.group(1), line.strip()))
            continue
        if name is None:
            continue
        if "s_endpgm" in line:
            name = None
        elif "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif not in_asm and not line.lstrip().startswith((";", ".")):
            for a, lo, hi in reg.findall(line.split(";")[0]):
                if (int(a) if a else int(hi)) >= row0:
                    bad.append((name, line.strip()))
    if len(kernels) != 3:          # generic, planar, rgb
        raise RuntimeError("tube2 audit: expected 3 stage_tube2_kernel instances, found %d" % len(kernels))
    if bad:
        raise RuntimeError("tube2 audit failed (the compiler touched the kernel's private registers, or the register budget changed): %r" % (bad[:5],))
    return len(kernels)


def build(force=False, verbose=False):
    """Compile the HIP extension in-tree for gfx950 (cross-compiles without a GPU).  The library is only put in place after
    audit_tube2_isa() has accepted the assembly of the same sources with the same flags."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libmulut_hip.so (no CPU fallback exists)")
    os.makedirs(_LIBDIR, exist_ok=True)
    tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())     # never expose a half-written library to another rank
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", tmp] + SOURCES
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd, cwd=_CSRC)
        audit_tube2_isa()
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


def load(path=None):
    """Return the ctypes handle; builds the library first if its sources are newer.
    `path` (or $MULUT_LIB) selects another build of the same ABI, e.g. an A/B kernel variant."""
    path = path or os.environ.get("MULUT_LIB") or LIB_PATH
    if path in _libs:
        return _libs[path]
    if path == LIB_PATH and needs_build():
        if os.environ.get("MULUT_NO_BUILD") == "1":
            # profiler runs (the preloaded tool has initialised the GPU) and non-zero ranks must never compile
            if not os.path.exists(LIB_PATH):
                raise RuntimeError("libmulut_hip.so is missing and MULUT_NO_BUILD=1 forbids building it here")
        elif _hipcc() is not None:
            build()
        elif not os.path.exists(LIB_PATH):
            raise RuntimeError("libmulut_hip.so is missing and hipcc is unavailable; "
                               "run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = ctypes.CDLL(path)
    i, p, c_char_p, i64 = ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64
    L.mulut_version.restype = i
    L.mulut_strerror.argtypes = [i]
    L.mulut_strerror.restype = c_char_p
    L.mulut_last_hip_error.argtypes = [p]
    L.mulut_last_hip_error.restype = c_char_p
    L.mulut_create.argtypes = [i, ctypes.POINTER(p)]
    L.mulut_destroy.argtypes = [p]
    L.mulut_configure.argtypes = [p, i, c_char_p, i, i]
    L.mulut_set_lut.argtypes = [p, i, ctypes.c_char, p, i64, i]
    L.mulut_pass.argtypes = [p, i, ctypes.c_char, i, p, i, i, i, p, p]
    L.mulut_stage.argtypes = [p, i, p, i, p, i, i, i, i, i, p]
    L.mulut_pipeline.argtypes = [p, p, p, i, i, i, i, i, p]
    L.mulut_pipeline_rows.argtypes = [p, p, i, i, p, i, i, i, i, i, i, i, p]
    L.mulut_halo.argtypes = [p]
    L.mulut_reserve.argtypes = [p, i, i, i, i]
    L.mulut_set_stage_timing.argtypes = [p, i]
    L.mulut_last_stage_ms.argtypes = [p, ctypes.POINTER(ctypes.c_float), i]
    L.mulut_last_kernel_ms.argtypes = [p, ctypes.POINTER(ctypes.c_float), i]
    L.mulut_last_detail_counters.argtypes = [p, ctypes.POINTER(ctypes.c_uint32), i, p]
    L.mulut_last_detail_counters.restype = i
    L.mulut_debug_read.argtypes = [p, ctypes.POINTER(ctypes.c_uint64), i, i, p]
    L.mulut_debug_read.restype = i
    L.mulut_set_tuning.argtypes = [p, c_char_p, i]
    L.mulut_kernel_name.argtypes = [p, i]
    L.mulut_ft_stage_forward.argtypes = [i, p, c_char_p, i, i, p, i, i, i, i, p, p]
    L.mulut_ft_stage_backward.argtypes = [i, p, c_char_p, i, i, p, p, i, i, i, i, p, p, p]
    L.mulut_ft_stage_forward_mask.argtypes = [i, p, c_char_p, i, i, p, i, i, i, i, p, p, p]
    L.mulut_ft_stage_backward_mask.argtypes = [i, p, c_char_p, i, i, p, p, p, i, i, i, i, p, p, p]
    L.mulut_kernel_name.restype = c_char_p
    L.mulut_ft_quantize.argtypes = [i, p, p, i, ctypes.c_longlong, p]
    L.mulut_ft_quantize_backward.argtypes = [i, p, p, i, ctypes.c_longlong, p]
    L.mulut_eval_ws_doubles.argtypes = [i, i]
    L.mulut_eval_ws_doubles.restype = ctypes.c_longlong
    L.mulut_eval_y.argtypes = [i, p, p, i, i, i, p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_double), p]
    L.mulut_eval_y.restype = i
    for name in ("mulut_create", "mulut_destroy", "mulut_configure", "mulut_set_lut", "mulut_pass", "mulut_stage",
                 "mulut_pipeline", "mulut_pipeline_rows", "mulut_halo", "mulut_reserve", "mulut_set_stage_timing",
                 "mulut_last_stage_ms", "mulut_last_kernel_ms", "mulut_set_tuning", "mulut_ft_stage_forward", "mulut_ft_stage_backward",
                 "mulut_ft_quantize", "mulut_ft_quantize_backward", "mulut_ft_stage_forward_mask", "mulut_ft_stage_backward_mask"):
        getattr(L, name).restype = i
    _libs[path] = L
    return L
