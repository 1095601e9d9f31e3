"""Aggregate a rocprofv3 PC-sampling CSV: samples per source line / per instruction of one kernel, with the mean number of
active lanes.   pcs_aggregate.py <csv> <out prefix> [kernel substring]"""
import collections
import csv
import sys

src, out = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
rows = csv.reader(open(src, newline=""))
hdr = next(rows)
print("columns:", hdr)
col = {h: i for i, h in enumerate(hdr)}
def find(*names):
    for n in names:
        for h, i in col.items():
            if n.lower() in h.lower():
                return i
    return None
c_ins, c_cmt, c_exec, c_disp = find("Instruction"), find("Comment"), find("Exec"), find("Dispatch")
if c_ins is not None and c_cmt == c_ins:
    c_cmt = None
for h, i in col.items():
    if h.lower() == "instruction": c_ins = i
    if "comment" in h.lower(): c_cmt = i
by_line = collections.Counter(); lanes_line = collections.Counter()
by_ins = collections.Counter(); lanes_ins = collections.Counter()
by_disp = collections.Counter()
n = 0
head = []
for r in rows:
    if len(head) < 30: head.append(r)
    n += 1
    ins = r[c_ins] if c_ins is not None else "?"
    cmt = r[c_cmt] if c_cmt is not None else "?"
    pop = 0
    if c_exec is not None:
        try: pop = bin(int(r[c_exec], 0) if r[c_exec].startswith("0x") else int(r[c_exec])).count("1")
        except ValueError: pop = 0
    by_line[cmt] += 1; lanes_line[cmt] += pop
    key = (cmt, ins)
    by_ins[key] += 1; lanes_ins[key] += pop
    if c_disp is not None: by_disp[r[c_disp]] += 1
with open(out + "_lines.txt", "w") as f:
    f.write(f"samples {n}\n")
    for k, v in by_line.most_common(400):
        f.write(f"{v:9d} {100.0*v/max(n,1):6.2f}% lanes {lanes_line[k]/max(v,1):5.1f}  {k}\n")
with open(out + "_ins.txt", "w") as f:
    for k, v in by_ins.most_common(1500):
        f.write(f"{v:9d} {100.0*v/max(n,1):6.2f}% lanes {lanes_ins[k]/max(v,1):5.1f}  {k[0]} | {k[1]}\n")
with open(out + "_head.txt", "w") as f:
    f.write(",".join(hdr) + "\n")
    for r in head: f.write(",".join(r) + "\n")
    f.write("dispatches: " + repr(by_disp.most_common(20)) + "\n")
print("samples", n)
