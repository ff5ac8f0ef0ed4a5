import sys
sys.path.insert(0, "/root/repo")
from icebergs_amd import synthetic as S
from icebergs_amd.framework import Icebergs
grid, p, b = S.config_c2(n=10_000_000, seed=2)
ib = Icebergs(grid, p, capacity=len(b["lon"]))
ib.upload_bergs(b); ib.set_store_environment(False)
for s in range(20):
    ib.run(1); ib.sync()
    print(s, ib.last_redo_count())
ib.close()
