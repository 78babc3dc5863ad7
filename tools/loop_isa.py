"""Measurement aid (no GPU): the instruction mix of a kernel's hottest loop from the compiler's assembly, block by block, with the cycle
model DESIGN 3.1 validated against the per-tile stamps (a plain VALU instruction 4 SIMD cycles per wave, packed fp32 / 64-bit moves 8,
transcendentals 16).  usage: python tools/loop_isa.py <substring of the mangled kernel name> [min v_exp in loop=8] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pf3plat_amd import _lib  # noqa: E402

TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")


def cost(op):
    if op.startswith(TRANS):
        return 16
    if op.startswith("v_pk_") or op.startswith("v_mov_b64") or op.endswith("_f64") or op.startswith("v_lshl_add_u64") or op.startswith("v_lshlrev_b64"):
        return 8
    return 4


def main():
    want = sys.argv[1]
    min_exp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + sys.argv[3:]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([_lib.find_hipcc(), *flags, "--cuda-device-only", "-S", "-o", out, _lib.SRC], check=True, capture_output=True)
        lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur, depth = [], None, {}
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            cur = {"label": m.group(1), "ops": [], "loop": None}
            c = m.group(2) or ""
            mh = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", c)
            if mh:
                cur["loop"] = "." + "L" + mh.group(1)
                depth[cur["loop"]] = int(mh.group(2))
            elif "Parent Loop" in c or "=>This" in c:
                cur["loop"] = cur["label"]  # a loop header
            blocks.append(cur)
            continue
        t = l.strip()
        if cur is None or not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ops"].append(t.split()[0])
    loops = {}
    for b in blocks:
        if b["loop"]:
            loops.setdefault(b["loop"], []).append(b)
    best = None
    for h, bs in loops.items():
        nexp = sum(op.startswith("v_exp_f32") for b in bs for op in b["ops"])
        nops = sum(len(b["ops"]) for b in bs)
        if nexp >= min_exp and (best is None or depth.get(h, 1) > best[1]):  # the innermost such loop
            best = (h, depth.get(h, 1), bs)
    if best is None:
        raise SystemExit("no loop with that many v_exp_f32")
    h, _, bs = best
    print(f"kernel {lines[start].split(':')[0]}  loop header {h}: {len(bs)} blocks")
    tot = {"valu": 0, "pk": 0, "trans": 0, "cyc": 0, "lds": 0, "salu": 0, "vmem": 0}
    for b in bs:
        v = [op for op in b["ops"] if op.startswith("v_")]
        cyc = sum(cost(op) for op in v)
        pk = sum(cost(op) == 8 for op in v)
        tr = sum(cost(op) == 16 for op in v)
        lds = sum(op.startswith("ds_") for op in b["ops"])
        sal = sum(op.startswith("s_") for op in b["ops"])
        vm = sum(op.startswith(("global_", "buffer_", "flat_")) for op in b["ops"])
        print(f"  {b['label']:14s} VALU {len(v):4d} (8-cycle {pk:3d}, transcendental {tr:2d}) = {cyc:5d} cycles | LDS {lds:3d} | SALU {sal:3d} | VMEM {vm:2d}")
        for k, x in (("valu", len(v)), ("pk", pk), ("trans", tr), ("cyc", cyc), ("lds", lds), ("salu", sal), ("vmem", vm)):
            tot[k] += x
    print(f"  all blocks     VALU {tot['valu']:4d} (8-cycle {tot['pk']:3d}, transcendental {tot['trans']:2d}) = {tot['cyc']:5d} cycles | LDS {tot['lds']:3d} | SALU {tot['salu']:3d} | VMEM {tot['vmem']:2d}")


if __name__ == "__main__":
    main()
