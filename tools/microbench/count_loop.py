"""Count the instructions in each mfma_valu kernel's timed loop (from hipcc --save-temps ISA)."""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z1kILi(\d+)ELi(\d+)ELi(\d+)EEv\w*):.*?\n(.*?)s_endpgm', s, re.S | re.M):
    lines = m.group(5).split('\n')
    labels = {re.match(r'^(\.LBB\d+_\d+):', l.strip()).group(1): i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l.strip())}
    best = None
    for i, l in enumerate(lines):
        mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            if best is None or (i - labels[mm.group(1)]) > (best[1] - best[0]):
                best = (labels[mm.group(1)], i)
    if best is None:
        print(f"NV={m.group(2)} MODE={m.group(3)}: no loop found"); continue
    loop = lines[best[0]:best[1]]
    mf = sum('v_mfma' in l for l in loop)
    va = sum(bool(re.match(r'\s+v_(?!mfma)', l)) for l in loop)
    sa = sum(bool(re.match(r'\s+s_', l)) for l in loop)
    print(f"NV={m.group(2):>3} MODE={m.group(3)} VPG={m.group(4):>2}: loop has {mf} MFMA, {va} VALU, {sa} scalar")
