"""CPU model of a lane-serial LZ4 block encoder (one lane per 1 KiB sub-chunk, 64 lanes in lockstep sharing one hash
table, optionally the previous 64 KiB region's table as a second one) on the C3' values block: compressed size against
liblz4's greedy parse (the oracle's block_compress is byte-identical to LZ4_compress_default).  Decides whether the parse
is worth building on the device: default_encoder_parity bounds the page bytes at 1.05 x the reference's.
usage: python scripts/sim/lz4_lanes_sim.py [sub_chunk_bytes] [hash_bits] [use_old 0/1]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import workloads
from oracle import sbo

SUB = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
HB = int(sys.argv[2]) if len(sys.argv) > 2 else 12
USE_OLD = int(sys.argv[3]) if len(sys.argv) > 3 else 1
WAYS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
LONGEST = int(sys.argv[5]) if len(sys.argv) > 5 else 0
LANES = 64
REGION = SUB * LANES
MINMATCH, MFLIMIT, LASTLIT = 4, 12, 5

col = workloads.zipf_utf8(65536, 42)
data = bytes(col["values"][: int(col["offsets"][65536])])
n = len(data)
ref = len(sbo.block_compress(sbo.LZ4, np.frombuffer(data, np.uint8)))
print("block", n, "bytes; liblz4", ref, "(%.3f)" % (ref / n))

def rd4(p):
    return int.from_bytes(data[p:p + 4], "little")
def h4(v):
    return ((v * 2654435761) & 0xFFFFFFFF) >> (32 - HB)

def seq_bytes(lit, mlen):
    s = 1 + lit + 2
    if lit >= 15: s += 1 + (lit - 15) // 255
    m = mlen - 4
    if m >= 15: s += 1 + (m - 15) // 255
    return s

total = 0
pending_lit_start = 0   # literals not yet in a sequence (carried across sub-chunks, as the stitch pass does)
nseq = 0
t_old = {}
for r0 in range(0, n, REGION):
    t_new = {}
    # lanes in lockstep: state per lane
    lanes = []
    for l in range(LANES):
        s = r0 + l * SUB
        if s >= n: break
        lanes.append(dict(pos=s, end=min(n, s + SUB), anchor=s, seqs=[]))
    active = True
    while active:
        active = False
        for L in lanes:      # one "step" of every lane (the model of a lockstep wave)
            p = L["pos"]
            if p >= L["end"]: continue
            active = True
            if p + MFLIMIT > n or p + 4 > L["end"]:
                L["pos"] = L["end"]
                continue
            v = rd4(p); h = h4(v)
            cands = list(t_new.get(h, ())) + (list(t_old.get(h, ())) if USE_OLD else [])
            t_new[h] = ((p,) + t_new.get(h, ()))[:WAYS]
            lim = min(L["end"], n - LASTLIT)
            best = -1; m = 0
            for c in cands:
                if c >= 0 and c < p and p - c <= 65535 and rd4(c) == v:
                    mm = 4
                    while p + mm < lim and data[c + mm] == data[p + mm]: mm += 1
                    if mm > m: best, m = c, mm
                    if not LONGEST: break
            if best < 0:
                L["pos"] = p + 1
                continue
            L["seqs"].append((p, m))
            L["pos"] = p + m
    t_old = t_new
    for L in lanes:
        for (p, m) in L["seqs"]:
            total += seq_bytes(p - pending_lit_start, m)
            pending_lit_start = p + m
            nseq += 1
lit = n - pending_lit_start
total += 1 + lit + (1 + (lit - 15) // 255 if lit >= 15 else 0)
print("lane-serial: sub %d, 2^%d entries, old table %d, ways %d, longest %d -> %d bytes = %.4f x liblz4, %d sequences" % (SUB, HB, USE_OLD, WAYS, LONGEST, total, total / ref, nseq))
