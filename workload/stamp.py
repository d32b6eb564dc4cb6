"""Content stamp of the kernel sources: profiles/r*_{kernel_stats.csv,traffic.json,pmc.json} carry the stamp of the `csrc/` they were
measured on, and bench.py refuses to quote a committed profile figure whose stamp differs from the tree it runs in (VERDICT r4 item 6:
a kernel change without a fresh profile session must not leave stale numbers in the judged line).  A content hash, not a git hash: the GPU
box gets a snapshot without `.git`."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "whisper-vits-svc_amd", "csrc", "*")))
    files += [os.path.join(root, "whisper-vits-svc_amd", "build.py"), os.path.join(root, "include", "svcmi.h")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode() + b"\0")
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
