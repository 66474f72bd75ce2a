# C3 sample buffers for the scripts/micro probes: 64 KiB of one column's values and offsets, and one page's values
# buffer (65 536 rows) with its LZ4 block as the oracle (liblz4's format, greedy) writes it
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import workloads as W
from oracle import sbo
c = W.zipf_utf8(200_000, 42)
c["values"][:65536].tofile("/tmp/c3_values.bin")
c["offsets"][:16384].astype(np.int32).tofile("/tmp/c3_offsets.bin")
page = c["values"][: int(c["offsets"][65536])]
page.tofile("/tmp/c3_page.bin")
blk = bytes(sbo.block_compress(sbo.LZ4, page))
open("/tmp/c3_page.lz4", "wb").write(blk)
print(len(page), len(blk))
