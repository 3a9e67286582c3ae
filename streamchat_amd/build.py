"""In-tree build of libstreamchat_hip.so (gfx950).  `python -m streamchat_amd.build`.

hipcc cross-compiles without a GPU, so this also is the CPU-side "does it build" check driven
by __graft_entry__.build().  The .so lands next to this file and travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstreamchat_hip.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(HERE, "..", "include", "streamchat_hip.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, jobs: int = 0) -> str:
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    if force or needs_build():
        j = jobs or (os.cpu_count() or 4)
        subprocess.check_call(["make", "-s", f"-j{j}", "-C", CSRC])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
