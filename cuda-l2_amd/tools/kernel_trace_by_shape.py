"""Per-shape kernel list from a rocprofv3 --kernel-trace of `hgemm_tune bench --shapes a,b,c ...` (ours or a baseline): the
operand-fill kernels of the bench separate the shapes in dispatch order.  Prints, per shape, every distinct kernel with its
average duration, grid, workgroup size, LDS and register counts, and the gap between the kernels of a two-kernel launch.

    python tools/kernel_trace_by_shape.py TRACE_DIR M_N_K,M_N_K,...

Used for profiles/r03_hipblaslt_kernels_on_loser_shapes.txt (what the vendor library runs where we lose)."""
import csv, glob, sys, re, collections
root, shapes = sys.argv[1], sys.argv[2].split(",")
f = glob.glob(f"{root}/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
groups, cur, in_fill = [], [], True
for r in rows:
    n = r["Kernel_Name"]
    setup = any(t in n for t in ("fill_normal", "transpose_kernel", "fillBuffer", "fill_"))
    if setup:
        if cur and not in_fill: groups.append(cur); cur = []
        in_fill = True; continue
    in_fill = False; cur.append(r)
if cur: groups.append(cur)
print(len(groups), "groups for", len(shapes), "shapes")
for s, g in zip(shapes, groups):
    names = []
    for r in g:
        if r["Kernel_Name"] not in names: names.append(r["Kernel_Name"])
    kpl = len(names); launches = len(g) // kpl
    print(f"== {s}: {kpl} kernel(s)/launch, {launches} launches")
    for n in names:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g if r["Kernel_Name"] == n][2:]
        r0 = [r for r in g if r["Kernel_Name"] == n][0]
        print(f"   {sum(d)/max(1,len(d))/1e3:8.2f} us  grid {r0.get('Grid_Size_X','?')}x{r0.get('Grid_Size_Y','?')}x{r0.get('Grid_Size_Z','?')} wg {r0.get('Workgroup_Size_X','?')} lds {r0.get('LDS_Block_Size','?')} vgpr {r0.get('VGPR_Count','?')} agpr {r0.get('Accum_VGPR_Count','?')}  {n[:400]}")
    if kpl > 1:
        # gap between consecutive kernels of one launch
        gaps = []
        for a, b in zip(g, g[1:]):
            if a["Kernel_Name"] != b["Kernel_Name"] and a["Kernel_Name"] == names[0]:
                gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
        if gaps: print(f"   gap first->second kernel: {sum(gaps)/len(gaps)/1e3:.2f} us")
