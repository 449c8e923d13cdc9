#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the HIP sources + the C-ABI header, sorted by file name.  ONE definition, two users: the
Makefile bakes it into libfcdgan_hip.so at build time (build/build_hash.h -> fcd_build_hash()), and _lib.kernel_source_hash()
evaluates it on the sources lying next to the package.  A committed PMC measurement (profiles/r*_hbm_traffic.json) is stamped
with the hash of the BINARY it was taken on, and bench.py reports it only while the loaded binary carries the same stamp."""
import glob
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def source_files():
    files = glob.glob(os.path.join(HERE, '*.hip')) + glob.glob(os.path.join(HERE, '*.h')) + \
        glob.glob(os.path.join(HERE, '..', '..', 'include', '*.h'))
    return sorted(files, key=os.path.basename)


def source_hash():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    out = '#define FCD_BUILD_HASH "%s"\n' % source_hash()
    if len(sys.argv) > 1:
        old = open(sys.argv[1]).read() if os.path.exists(sys.argv[1]) else None
        if old != out:
            with open(sys.argv[1], 'w') as f:
                f.write(out)
    else:
        sys.stdout.write(out)
