"""Measurement aid: build variants of the library with -D switches into tools/variants/<name>.so (cross-compiles here;
the files travel to the GPU box).  usage: python tools/build_variants.py name=-DA=1,-DB=2 ..."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import _lib  # noqa: E402

os.makedirs(os.path.join(ROOT, "tools", "variants"), exist_ok=True)


def one(spec):
    name, _, flags = spec.partition("=")
    out = os.path.join(ROOT, "tools", "variants", f"{name}.so")
    _lib.build(force=True, extra_flags=[f for f in flags.split(",") if f], out=out)
    return out


with ThreadPoolExecutor(8) as ex:
    for o in ex.map(one, sys.argv[1:]):
        print("built", o)
