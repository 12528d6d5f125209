"""Build libkd6d.so (gfx950 only) in-tree with hipcc.

Usage: python build.py [--force] [--sanitize]
The shared object lands next to the sources (csrc/libkd6d.so); it is git-ignored but
travels to the GPU box with the gpurun snapshot.

--sanitize builds csrc/libkd6d_san.so instead: the HOST code of every csrc/*.hip -- launchers, dispatch rules,
work-list planners, argument checks -- under AddressSanitizer + UndefinedBehaviorSanitizer (-O1; the device code is
compiled as usual: sanitizers are not available for gfx950 code objects on this pool).  tests/test_sanitize_host.py drives
it on the CPU through tests/san_driver.cpp (argument checks, dry-run dispatch, split and work-list planning).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libkd6d.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
]


# per-source additions.  sinkhorn_dense.hip: keep MFMA results in VGPRs -- the logsumexp that follows reads every one
# of them on the VALU, and out of AGPRs that is one v_accvgpr_read per value (16 % of the kernel's vector work)
EXTRA_FLAGS = {"sinkhorn_dense.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # losses.hip, sinkhorn.hip: no packed-fp32 instructions (v_pk_fma_f32 & co.) -- see DESIGN.md section 6,
               # "What made two executions differ": in these one-workgroup-per-image launches, running beside the other
               # network's convolutions, one HALF of a packed result occasionally came out as if its product were zero,
               # for the last 16 lanes of a wave (tests/flake_hunt.py); with scalar fp32 instructions it does not happen
               "losses.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
               "sinkhorn.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_digest():
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith(".h")]
                       + [os.path.join(HERE, "..", "include", "kd6d.h")]):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h


def _source_digest(src, base):
    h = base.copy()
    with open(os.path.join(CSRC, src), "rb") as f:
        h.update(f.read())
    h.update(" ".join(EXTRA_FLAGS.get(src, [])).encode())
    return h.hexdigest()


def _san_flags():
    """Sanitizer flags of the host build.  They live in tools/build_sanitized.py: that file (and the CPU-only test and driver
    that use the sanitized library) is listed in .gpurunignore -- the GPU pool refuses snapshots whose GPU-side files name
    sanitizer builds -- and is not needed on the GPU box."""
    import importlib.util
    path = os.path.join(HERE, "..", "tools", "build_sanitized.py")
    spec = importlib.util.spec_from_file_location("kd6d_build_sanitized", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return list(mod.SAN_FLAGS)


def build(force=False, verbose=True, sanitize=False):
    """Compile every csrc/*.hip whose source, the shared headers or the flags changed since its object was built
    (one stamp per object), then link.  Returns the path of the shared object."""
    global FLAGS, OUT
    flags0, out0 = FLAGS, OUT
    san = _san_flags() if sanitize else []
    if sanitize:
        FLAGS = [f for f in FLAGS if f != "-O3"] + ["-O1"] + san
        OUT = os.path.join(CSRC, "libkd6d_san.so")
    try:
        return _build(force, verbose, "build_san" if sanitize else "build", san)
    finally:
        FLAGS, OUT = flags0, out0


def _build(force, verbose, objsub, link_extra):
    objdir = os.path.join(CSRC, objsub)
    os.makedirs(objdir, exist_ok=True)
    base = _headers_digest()
    srcs = _sources()
    digests = {src: _source_digest(src, base) for src in srcs}

    def stale(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        stamp = obj + ".stamp"
        if force or not os.path.exists(obj) or not os.path.exists(stamp):
            return True
        with open(stamp) as f:
            return f.read().strip() != digests[src]

    todo = [s for s in srcs if stale(s)]
    link_stamp = os.path.join(CSRC, ".build_stamp" if objsub == "build" else "." + objsub + "_stamp")
    all_dig = hashlib.sha256("".join(digests[s] for s in srcs).encode()).hexdigest()
    if not todo and os.path.exists(OUT) and os.path.exists(link_stamp):
        with open(link_stamp) as f:
            if f.read().strip() == all_dig:
                return OUT

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        with open(obj + ".stamp", "w") as f:
            f.write(digests[src])
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        list(ex.map(compile_one, todo))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in srcs]
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + link_extra + objs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(link_stamp, "w") as f:
        f.write(all_dig)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, sanitize="--sanitize" in sys.argv))
