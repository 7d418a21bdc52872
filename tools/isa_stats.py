"""Developer tool (CPU): static instruction statistics of one kernel in a hipcc -S listing -- size, SGPR-spill traffic (v_readlane / v_writelane /
hazard s_nop), f64 divisions, DPP and LDS instructions, register metadata.  `python tools/isa_stats.py listing.s solve_kernelIfLi5`."""
import collections
import re
import sys


def kernel_body(text, key):
    m = re.search(r"^(_Z\w*%s\w*):" % re.escape(key), text, re.M)
    if not m:
        raise SystemExit(f"no kernel matching {key}")
    name = m.group(1)
    end = text.index(".size\t" + name, m.end()) if (".size\t" + name) in text else len(text)
    return name, text[m.end():end]


def main():
    text = open(sys.argv[1]).read()
    for key in sys.argv[2:]:
        name, body = kernel_body(text, key)
        ins = [l.strip() for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
        c = collections.Counter(l.split()[0] for l in ins)
        meta = {}
        i = text.index("amdhsa.kernels:")
        for blk in text[i:].split("  - .agpr_count:")[1:]:
            if name in blk:
                meta["agpr"] = blk.split("\n")[0].strip()
                for k in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size"):
                    mm = re.search(r"\.%s:\s+(\S+)" % k, blk)
                    meta[k] = mm.group(1) if mm else None
        grp = lambda pred: sum(v for k, v in c.items() if pred(k))
        print(f"{key}: {len(ins)} instructions; readlane {c['v_readlane_b32']} writelane {c['v_writelane_b32']} s_nop {c['s_nop']} "
              f"f64 div {c['v_div_fixup_f64']} dpp {grp(lambda k: 'dpp' in k)} lds {grp(lambda k: k.startswith('ds_'))} "
              f"scratch {grp(lambda k: k.startswith('scratch_'))} salu {grp(lambda k: k.startswith('s_'))} cndmask {grp(lambda k: k.startswith('v_cndmask'))}")
        print("   ", meta)
        if "--top" in sys.argv:
            for k, v in c.most_common(30):
                print(f"    {v:6d} {k}")


if __name__ == "__main__":
    main()
