"""Derived view of tools/pmc_table.py's raw SQ-counter table (markdown in, markdown out): per kernel, instructions per wave and the
shares that say what bounds it.  usage: python tools/sq_derived.py <raw table.md> > derived.md"""
import sys

rows = [ln.strip().strip("|").split("|") for ln in open(sys.argv[1]) if ln.startswith("|")]
hdr = [h.strip() for h in rows[0]]
print("| kernel | waves | VALU / wave | of which transcendental | LDS / wave | SALU / wave | VMEM rd / wr per wave | VALU-active share of wave-cycles | "
      "waiting-on-instruction share | LDS bank-conflict share of LDS-active |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows[2:]:
    d = {h: (float(v) if v.strip() not in ("",) and h != "kernel" else v.strip()) for h, v in zip(hdr, r)}
    w = d["SQ_WAVES"]
    if w < 64:
        continue
    print(f"| {d['kernel']} | {w:.0f} | {d['SQ_INSTS_VALU'] / w:.0f} | {d['SQ_INSTS_VALU_TRANS_F32'] / w:.0f} | {d['SQ_INSTS_LDS'] / w:.0f} | "
          f"{d['SQ_INSTS_SALU'] / w:.0f} | {d['SQ_INSTS_VMEM_RD'] / w:.1f} / {d['SQ_INSTS_VMEM_WR'] / w:.1f} | "
          f"{d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES']:.2f} | {d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:.2f} | "
          f"{d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1):.2f} |")
