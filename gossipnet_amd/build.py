"""Builds libgossipnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# measurement builds (probe variants of a kernel, workgroup time stamps) never overwrite the shipped library: they are
# written to -- and, by a process with the same environment, loaded from -- a library of their own
PROBE_BUILD = bool(os.environ.get("GNET_EXTRA_FLAGS") or os.environ.get("GNET_TRACE"))
LIB = os.path.join(HERE, "libgossipnet_hip_probe.so" if PROBE_BUILD else "libgossipnet_hip.so")
SOURCES = ["graph.hip", "forward.hip", "loss.hip", "backward.hip", "backward_edge.hip", "roi_pool.hip", "optim.hip", "fc.hip", "plan.hip", "debug.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]
if os.environ.get("GNET_EXTRA_FLAGS"):    # measurement builds only (probe variants of a kernel: tools/); never shipped
    FLAGS = FLAGS + os.environ["GNET_EXTRA_FLAGS"].split()
if os.environ.get("GNET_TRACE"):          # measurement build with per-workgroup time stamps (tools/wg_trace.py); never shipped
    FLAGS = FLAGS + ["-DGNET_TRACE"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_hash():
    """Identity of the sources a library is built from (written next to the .so; _lib.load() checks it)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    for f in files:
        h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "gossipnet_hip.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _up_to_date():
    try:
        return os.path.exists(LIB) and open(LIB + ".srchash").read().strip() == source_hash()
    except OSError:
        return False


def build(force=False, verbose=False):
    """Compiles what changed and links the library.  Concurrent callers (one process per GPU) are serialised by an
    exclusive lock on csrc/build/.lock; a caller that waited finds the library current and returns."""
    import fcntl
    import hashlib
    # objects of different flag sets (the GNET_TRACE measurement build) never mix: one directory per flag set
    objdir = os.path.join(CSRC, "build", hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()[:8])
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(CSRC, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date():
                return LIB
            return _build_locked(force, verbose, objdir)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, objdir):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "backward_edge.hpp"),
                   os.path.join(HERE, "..", "include", "gossipnet_hip.h")]
    hdr_mtime = max(os.path.getmtime(d) for d in deps[len(srcs):])
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_mtime):
            cmd = [_hipcc()] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out):
            sys.stderr.write(out.decode())
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    # (reached only when the library's recorded hash differs from this source / flag set: always link -- the objects of
    # this flag set may all be current while the library was linked from another set's)
    tmp = LIB + ".tmp.%d" % os.getpid()
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
    os.replace(tmp, LIB)                     # a process that has the old library mapped keeps its inode
    with open(LIB + ".srchash", "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
