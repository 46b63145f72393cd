#!/usr/bin/env python3
"""Pack the reference's test corpus (test/corpus/*: the 14 Calgary files + rfc5322.txt, the data SURVEY.md 8(d)
builds configs C2 and C4 from) into ONE data fixture, tests/golden/corpus.tar.xz.

Run HERE (the build container, where /root/reference exists):   python tests/golden/gen_corpus.py
The files are data, not source; they travel to the GPU box inside the fixture."""
import io
import lzma
import os
import tarfile

REF = "/root/reference/test/corpus"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "corpus.tar.xz")


def main():
    names = sorted(os.listdir(REF))
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w") as tar:
        for n in names:
            ti = tar.gettarinfo(os.path.join(REF, n), arcname=n)
            ti.uid = ti.gid = 0
            ti.uname = ti.gname = ""
            ti.mtime = 0
            with open(os.path.join(REF, n), "rb") as f:
                tar.addfile(ti, f)
    with open(OUT, "wb") as f:
        f.write(lzma.compress(buf.getvalue(), preset=9 | lzma.PRESET_EXTREME))
    total = sum(os.path.getsize(os.path.join(REF, n)) for n in names)
    print("%s: %d files, %d bytes -> %d bytes" % (OUT, len(names), total, os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
