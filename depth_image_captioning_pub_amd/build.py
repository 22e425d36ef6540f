"""Build libdic_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
Usage: python -m depth_image_captioning_pub_amd.build [--force] [--experiments]

--experiments builds a SECOND library, libdic_experiments.so: the same sources compiled with -DDIC_EXPERIMENTS plus
csrc/experiments/*.hip - the parked kernels (persistent decoder loop, pipe / 256x128 contraction forms), the packed-fp32
defect reproducer and every benchmarking / ablation switch.  scripts/ use it (DIC_LIB=experiments); the product library,
the tests and bench.py never load it.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
OBJ_EXP = os.path.join(CSRC, "_obj", "experiments")
LIB = os.path.join(HERE, "libdic_hip.so")
LIB_EXP = os.path.join(HERE, "libdic_experiments.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-save-temps=obj"]      # (the device assembly is audited below)

# Packed fp32 instructions whose LOW result takes the HIGH half of src1 (op_sel bit 1) give wrong results on gfx950 when the wave
# shares its SIMD with other kernels' waves: 900 000 wrong sums of 1.6e10 for `v_pk_add_f32 ... op_sel:[0,1]` next to a concurrent
# bf16x3 forward, none alone, none for any other operand-select form (scripts/diag_pk_fp32_opsel.py; found through
# csrc/conv1_depth.hip, see its header).  hipcc chooses these forms by itself (float2 broadcasts), so every build audits the device
# assembly it has just produced.  Nothing linked into libdic_hip.so is exempt.  The experiments library holds two sources that
# contain the bad forms - the reproducer of the defect itself and the parked persistent decoder loop (do not run that one
# next to other streams) - and exempts exactly those.
_AUDIT_EXEMPT = ()
_AUDIT_EXEMPT_EXPERIMENTS = ("decoder_persist", "probe_pk_fp32")
_PK_F32 = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b.*\bop_sel:\[([01]),([01])")


def audit_packed_fp32(verbose: bool = False, obj_dir: str = OBJ) -> dict:
    """{source stem: number of packed fp32 instructions with op_sel[1] = 1} over the device assembly in csrc/_obj."""
    found = {}
    for f in sorted(os.listdir(obj_dir)):
        if not f.endswith(f"-hip-amdgcn-amd-amdhsa-{ARCH}.s"):
            continue
        stem = f.split("-hip-")[0]
        n = 0
        with open(os.path.join(obj_dir, f)) as fh:
            for line in fh:
                m = _PK_F32.match(line)
                if m and m.group(3) == "1":
                    n += 1
                    if verbose:
                        sys.stderr.write(f"{stem}: {line.strip()}\n")
        found[stem] = n
    return found


_SPILL = re.compile(r"^\s*\.vgpr_spill_count:\s*(\d+)")
_KSYM = re.compile(r"^\s*\.symbol:\s*(\S+?)(?:\.kd)?\s*$")


def audit_register_spills(obj_dir: str = OBJ) -> dict:
    """{kernel symbol: spilled vector registers} over the kernel metadata of the device assembly.  The hand-scheduled kernels issue loads
    through inline asm and count `vmcnt` themselves: a register the compiler spills while such a load is in flight is stored before its
    data has arrived, and the spill traffic itself joins the queue the kernel counts (round 4: an eight-slot variant of the
    on-the-fly-operand kernel that spilled 161 registers hung on the device).  No kernel of the product library may spill vector registers."""
    found = {}
    for f in sorted(os.listdir(obj_dir)):
        if not f.endswith(f"-hip-amdgcn-amd-amdhsa-{ARCH}.s"):
            continue
        sym = None
        with open(os.path.join(obj_dir, f)) as fh:
            for line in fh:
                if line.lstrip().startswith("- .agpr_count"):        # a new entry of amdhsa.kernels
                    sym = None
                m = _KSYM.match(line)
                if m:
                    sym = m.group(1)
                m = _SPILL.match(line)
                if m and int(m.group(1)) > 0:
                    found[sym or f"{f}:?"] = int(m.group(1))
    return found


def _hipcc() -> str:
    h = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(h):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return h


def _sources(experiments: bool = False):
    src = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]
    if experiments:
        exp = os.path.join(CSRC, "experiments")
        src += [os.path.join(exp, f) for f in sorted(os.listdir(exp)) if f.endswith(".hip")]
    return src


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    obj_dir, lib_path = (OBJ_EXP, LIB_EXP) if experiments else (OBJ, LIB)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "dic.h"))
    if experiments:
        exp = os.path.join(CSRC, "experiments")
        headers += [os.path.join(exp, f) for f in os.listdir(exp) if f.endswith((".h", ".inc"))]
    headers = [h for h in headers if os.path.exists(h)]
    jobs = []
    objs = []
    for s in _sources(experiments):
        o = os.path.join(obj_dir, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + (["-DDIC_EXPERIMENTS"] if experiments else []) + ["-I", CSRC, "-I", os.path.join(os.path.dirname(HERE), "include")]
            if s.endswith(".cpp"):
                cmd += ["-x", "hip"]
            cmd += ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or r.returncode != 0:
            sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for " + cmd[-3])
        return 0

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    for f in os.listdir(obj_dir):      # -save-temps leaves ~9 MB of intermediates per source; only the device assembly is of interest
        if f.endswith((".hipi", ".bc", ".hipfb", ".resolution.txt", ".out")) or "-host-" in f or f.endswith(f"{ARCH}.o"):
            os.remove(os.path.join(obj_dir, f))
    stems = {os.path.basename(o)[:-2] for o in objs}
    for f in os.listdir(obj_dir):      # objects / listings of sources that no longer belong to this library
        stem = f.split("-hip-")[0] if "-hip-" in f else f.rsplit(".", 1)[0]
        if os.path.isfile(os.path.join(obj_dir, f)) and stem not in stems:
            os.remove(os.path.join(obj_dir, f))
    exempt = _AUDIT_EXEMPT_EXPERIMENTS if experiments else _AUDIT_EXEMPT
    bad = {k: v for k, v in audit_packed_fp32(obj_dir=obj_dir).items() if v and k not in exempt}
    if bad:
        raise RuntimeError(f"packed fp32 instructions with op_sel[1] = 1 in the device code of {bad}: see the note in build.py "
                           "(python -c 'from depth_image_captioning_pub_amd import build; build.audit_packed_fp32(True)' lists them)")
    # (exempt by name: conv1_wgrad_kernel - the depth encoder's layer-1 weight gradient, 2..34 spilled registers at three waves per SIMD,
    #  every load compiler-visible, no hand-counted wait in that kernel: a speed matter only)
    spills = {k: v for k, v in audit_register_spills(obj_dir).items() if "conv1_wgrad_kernel" not in k}
    if spills and not experiments:
        raise RuntimeError(f"kernels that spill vector registers (see audit_register_spills): {spills}")
    if jobs or force or _stale(lib_path, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path] + [o for o in objs])
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, experiments="--experiments" in sys.argv))
