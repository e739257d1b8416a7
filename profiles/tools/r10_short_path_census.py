"""Round 6 (no GPU): a static census of the cycle kernel's short cycle.  Input: the gfx950 assembly of zkw_kernels.hip built with
-DZKW_ASM_MARKS (the marks `loop top`, `short test`, `short qualified`, `short record`, `short end` of the source).  The kernel's
control-flow graph is rebuilt from the labels and branches, and the CHEAPEST route (fewest instructions) loop top -> short test ->
short qualified -> short record -> short end -> loop top is printed by instruction class: a lower bound of what one short cycle
issues (a NOP takes no other route through the qualified region: no operand, no access, no register write), and the list of what
that bound consists of.   hipcc ... -DZKW_ASM_MARKS -S --cuda-device-only -o k.s zkw_kernels.hip; python r10_short_path_census.py k.s"""
import collections, heapq, re, sys

src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if re.match(r"^_Z16zkw_cycle_kernel\w*:", l))
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
blocks, order, cur = {}, [], None
marks = {}
def new_block(name):
    global cur
    cur = name
    blocks[name] = {"ins": [], "succ": [], "fall": True}
    order.append(name)
new_block("entry")
anon = 0
for i in range(start + 1, end):
    l = src[i]
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        new_block(m.group(1)); continue
    if s.startswith("; %bb."):
        new_block("bb%d_%s" % (i, s.split()[1])); continue
    mm = re.match(r"^; MARK (.*)$", s)
    if mm:
        name = "MARK:" + mm.group(1)
        new_block(name); marks[mm.group(1)] = name; continue
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    op = s.split()[0]
    blocks[cur]["ins"].append(s)
    if op.startswith("s_cbranch"):
        blocks[cur]["succ"].append(s.split()[1]); anon += 1; new_block("ft%d" % anon)
    elif op == "s_branch":
        blocks[cur]["succ"].append(s.split()[1]); blocks[cur]["fall"] = False; anon += 1; new_block("dead%d" % anon)
    elif op in ("s_endpgm", "s_setpc_b64"):
        blocks[cur]["fall"] = False; anon += 1; new_block("dead%d" % anon)
for k, name in enumerate(order[:-1]):
    if blocks[name]["fall"]:
        blocks[name]["succ"].append(order[k + 1])

def cheapest(a, b):
    """fewest instructions from the start of block a to the start of block b"""
    dist, prev = {a: 0}, {}
    pq = [(0, a)]
    while pq:
        d, u = heapq.heappop(pq)
        if u == b:
            break
        if d > dist.get(u, 1 << 60):
            continue
        for v in blocks[u]["succ"]:
            if v not in blocks:
                continue
            nd = d + len(blocks[u]["ins"])
            if nd < dist.get(v, 1 << 60):
                dist[v] = nd; prev[v] = u; heapq.heappush(pq, (nd, v))
    path, u = [], b
    while u != a:
        u = prev[u]; path.append(u)
    return dist[b], path[::-1]

def klass(s):
    op = s.split()[0]
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): return "lane <-> scalar moves (v_readlane / v_writelane / v_readfirstlane)"
    if op == "s_nop": return "s_nop"
    if op == "s_waitcnt": return "s_waitcnt"
    if op.startswith("s_cbranch") or op == "s_branch": return "branches"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "scalar loads"
    if op.startswith("s_"): return "other SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vector memory"
    if op.startswith("v_"): return "VALU"
    return "other"

route = ["loop top", "short test", "short qualified", "short record", "short end", "loop top"]
if len(sys.argv) > 2 and sys.argv[2] == "--access":  # the cheapest heap access instead (through the marks around its read queries): an aligned read
    route = ["loop top", "short test", "short qualified", "emit0 begin", "emit0 end", "short record", "short end", "loop top"]
    sys.argv = sys.argv[:2]
total = collections.Counter(); n_total = 0
print("cheapest route through the short cycle, by leg (instructions):")
for a, b in zip(route, route[1:]):
    d, path = cheapest(marks[a], marks[b])
    c = collections.Counter()
    for u in path:
        for s in blocks[u]["ins"]:
            c[klass(s)] += 1
    total += c; n_total += d
    print("  %-16s -> %-16s %4d   %s" % (a, b, d, ", ".join("%s %d" % (k.split(" (")[0], v) for k, v in c.most_common())))
print("whole cycle: %d instructions" % n_total)
for k, v in total.most_common():
    print("  %4d  %s" % (v, k))
if len(sys.argv) > 2:  # list one leg: python r10_short_path_census.py k.s "short test"
    a = sys.argv[2]; b = route[route.index(a) + 1]
    d, path = cheapest(marks[a], marks[b])
    for u in path:
        print("%s:" % u)
        for s in blocks[u]["ins"]:
            print("\t" + s)
