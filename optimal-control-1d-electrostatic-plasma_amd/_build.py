"""Build libpicstep.so for gfx950 with hipcc, in-tree (csrc/libpicstep.so).

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "picstep.hip")
LIB = os.path.join(HERE, "csrc", "libpicstep.so")
RESOURCES = os.path.join(HERE, "csrc", "libpicstep.resources.json")   # registers / scratch / LDS of every kernel in LIB
INCLUDE = os.path.join(ROOT, "include")

# -ffp-contract=off: the sub-stage arithmetic must round like the NumPy reference (no FMA fusion)
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-pthread",
         "-Wno-unused-result", "-Wno-unused-value", "-I" + INCLUDE, "-Rpass-analysis=kernel-resource-usage"]


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
        return True
    t = os.path.getmtime(LIB)
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(INCLUDE, "picstep.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(out, defines=(), verbose=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libpicstep.so can only be built with the ROCm toolchain")
    tmp = f"{out}.tmp.{os.getpid()}"           # several ranks may build at once: private file, atomic rename
    cmd = [hipcc] + FLAGS + ["-D" + d for d in defines] + ["-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, out)
    if out == LIB:
        _write_resources(r.stderr)
    return out


def _write_resources(remarks):
    """The compiler's per-kernel resource report next to the library: tests/test_host_cpu.py holds every kernel the host can
    launch to zero scratch (a register spill in a particle loop costs a factor, not a percentage, and no parity test sees it)."""
    import json
    import re
    kernels = {}
    for block in re.split(r"remark: [^\n]*Function Name: ", remarks)[1:]:
        name = block.split()[0]

        def field(key):
            m = re.search(key + r": (\d+)", block)
            return int(m.group(1)) if m else None
        kernels[name] = {"vgprs": field("VGPRs"), "agprs": field("AGPRs"), "sgprs": field("SGPRs"),
                         "scratch_bytes_per_lane": field(r"ScratchSize \[bytes/lane\]"),
                         "lds_bytes_per_block": field(r"LDS Size \[bytes/block\]"), "waves_per_simd": field(r"Occupancy \[waves/SIMD\]")}
    tmp = f"{RESOURCES}.tmp.{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump(kernels, f, indent=0, sort_keys=True)
    os.replace(tmp, RESOURCES)


def build_library(force=False, verbose=False):
    """Compile csrc/picstep.hip (+ its pic_*.h kernel headers) -> csrc/libpicstep.so. Returns the library path."""
    if not force and not needs_build():
        return LIB
    return _compile(LIB, (), verbose)
