"""Dev probe (GPU box): where a pass of the statistics chunk driver spends its time -- chunk loop (host wall / GPU events) vs merge."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import syconn_amd.proc.sd_proc as sp
from syconn_amd.extraction.find_object_properties import DeviceScan
from tools.propmerge_bench import synth
dev = torch.device('cuda', 0)
ext, cs = np.array([2048, 2048, 512]), np.array([512, 512, 512])
names = ['mi', 'vc', 'sj']
store = {}
origins = [np.array([x, y, 0]) for x in range(0, 2048, 512) for y in range(0, 2048, 512)]
for o in origins:
    for n in ['sv'] + names:
        store[(n, tuple(o))] = synth(n, o, cs, dev)
mov = {'sv': 1, 'mi': 200, 'vc': 200, 'sj': 200}
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    scan = DeviceScan(dev, 1 << 19, 1 << 19)
    merger = sp.ChunkMerger(names, mov, dev, len(origins))
    t_a = time.perf_counter(); e0.record()
    marks = []
    for o in origins:
        t1 = time.perf_counter()
        scan.scan(store[('sv', tuple(o))], [store[(n, tuple(o))] for n in names], status_out=merger.status_slot())
        t2 = time.perf_counter()
        merger.add_chunk(scan, o)
        marks.append((t2 - t1, time.perf_counter() - t2))
    e1.record(); t_b = time.perf_counter()
    res = merger.finish()
    e2.record(); torch.cuda.synchronize(); t_c = time.perf_counter()
    print(f'rep {rep}: setup {1e3 * (t_a - t0):.1f} ms | loop host {1e3 * (t_b - t_a):.1f} ms, GPU {e0.elapsed_time(e1):.1f} ms | finish host {1e3 * (t_c - t_b):.1f} ms, GPU {e1.elapsed_time(e2):.1f} ms'
          f' | per chunk host: scan {1e3 * np.mean([m[0] for m in marks]):.2f} add {1e3 * np.mean([m[1] for m in marks]):.2f} ms (max {1e3 * max(m[0] + m[1] for m in marks):.2f})')
