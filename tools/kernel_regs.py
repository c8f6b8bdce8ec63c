"""Print VGPR/AGPR/spill/scratch/LDS of every kernel in a hipcc -S listing (tools, not product)."""
import re, sys
for path in sys.argv[1:]:
    txt = open(path).read()
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        ag, name, priv, ss, vg, vs = m.groups()
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:60]
        print(f"{name:62s} vgpr {vg:>3s} agpr {ag:>3s} vspill {vs:>3s} sspill {ss:>3s} scratch {priv}")
