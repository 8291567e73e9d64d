"""Hash of the native sources (csrc/*.hip, *.h, *.cpp, *.c, Makefile + include/upamd.h): stamps the committed PMC summaries so
bench.py can tell whether they still describe the kernels it is running."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    h = hashlib.sha1()
    base = os.path.join(root, 'drl-urban-planning_amd', 'csrc')
    files = sorted(f for pat in ('*.hip', '*.h', '*.cpp', '*.c', 'Makefile') for f in glob.glob(os.path.join(base, pat)))
    files.append(os.path.join(root, 'include', 'upamd.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


if __name__ == '__main__':
    print(csrc_hash())
