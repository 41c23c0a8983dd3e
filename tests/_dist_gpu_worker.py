"""Worker of tests/test_gpu_multirank.py (launched by torch.distributed.run, 2 ranks): the stream / event protocol of
``predict_volume_distributed`` ON THE DEVICE -- copy streams, communication stream, buffer-reuse events, scatter(r+1) issued before
predict(r), ``root_computes=False``, the pinned pool -- compared bit for bit with the single-process result for every combination
of `pipelined` and `root_computes`, with >= 5 rounds and a ragged last round.  Backend: RCCL when the box has two GPUs, otherwise
gloo with host-staged payloads and both ranks on cuda:0 (syconn_amd.parallel._staged).  Launched as ONE rank with
``SD_DIST_SINGLE_RANK_GROUP=1`` the same protocol runs through a process group of one on RCCL (device tensors in ``dist.scatter`` /
``dist.gather``, asynchronous work handles waited on from the compute / communication stream): the "nccl" branches a one-GPU box can
execute."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from syconn_amd import parallel as par
    single = os.environ.get('SD_DIST_SINGLE_RANK_GROUP') == '1' and int(os.environ.get('WORLD_SIZE', '1')) == 1
    two = torch.cuda.device_count() >= 2 or single
    rank, world, local_rank = par.init_distributed('nccl' if two else 'gloo')
    if single:
        assert torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl' and par._collectives()
        # Coll-1 on RCCL: the broadcast of the flat weight vector as a DEVICE tensor
        sd = {'a.weight': torch.arange(24, dtype=torch.float32).reshape(2, 3, 4), 'a.bias': torch.ones(2)}
        par.broadcast_weights(sd, src=0, device=torch.device('cuda', 0))
        assert torch.equal(sd['a.weight'], torch.arange(24, dtype=torch.float32).reshape(2, 3, 4))
        assert par.max_over_ranks(3.5, torch.device('cuda', 0)) == 3.5
    dev = torch.device('cuda', local_rank if two else 0)
    torch.cuda.set_device(dev)
    from syconn_amd.engine import require_gpu
    require_gpu(dev.index)
    vol_shape, chunk, halo = (40, 70, 66), (16, 24, 32), (2, 3, 4)      # grid 3 x 3 x 3 = 27 chunks: 14 / 27 rounds, ragged ends
    g = torch.Generator().manual_seed(5)
    vol = torch.randint(0, 256, vol_shape, dtype=torch.uint8, generator=g)
    k = torch.arange(1, 1 + (2 * halo[0] + 1), dtype=torch.float32)

    def predict_fn(ch):
        # an asynchronous device function that USES the halo: a separable-ish box statistic of the chunk + halo, two outputs
        x = ch.to(torch.int32)
        z0, y0, x0 = halo
        inner = x[z0:x.shape[0] - z0, y0:x.shape[1] - y0, x0:x.shape[2] - x0]
        up = x[:x.shape[0] - 2 * z0, y0:x.shape[1] - y0, x0:x.shape[2] - x0]
        left = x[z0:x.shape[0] - z0, y0:x.shape[1] - y0, 2 * x0:]
        a = ((inner * 3 + up + 2 * left) % 251).to(torch.uint8)
        b = ((inner + x[z0:x.shape[0] - z0, 2 * y0:, x0:x.shape[2] - x0] * 5) % 253).to(torch.uint8)
        for _ in range(3):                       # some queue depth on the compute stream
            a = (a.to(torch.int32) * 1).to(torch.uint8)
        return torch.stack([a, b])

    # single-process reference on rank 0's device (the same function over the zero-padded volume, chunk by chunk)
    want = None
    if rank == 0:
        pad = torch.zeros(tuple(v + 2 * h for v, h in zip(vol_shape, halo)), dtype=torch.uint8)
        pad[halo[0]:halo[0] + vol_shape[0], halo[1]:halo[1] + vol_shape[1], halo[2]:halo[2] + vol_shape[2]] = vol
        # chunks may overhang the volume: pad further with zeros up to the chunk grid
        grid = [-(-v // c) for v, c in zip(vol_shape, chunk)]
        big = torch.zeros(tuple(gd * c + 2 * h for gd, c, h in zip(grid, chunk, halo)), dtype=torch.uint8)
        big[:pad.shape[0], :pad.shape[1], :pad.shape[2]] = pad
        want = torch.zeros((2, *vol_shape), dtype=torch.uint8)
        for iz in range(grid[0]):
            for iy in range(grid[1]):
                for ix in range(grid[2]):
                    lo = (iz * chunk[0], iy * chunk[1], ix * chunk[2])
                    sub = big[lo[0]:lo[0] + chunk[0] + 2 * halo[0], lo[1]:lo[1] + chunk[1] + 2 * halo[1],
                              lo[2]:lo[2] + chunk[2] + 2 * halo[2]].to(dev)
                    r = predict_fn(sub).cpu()
                    n = [min(c, v - l) for c, v, l in zip(chunk, vol_shape, lo)]
                    want[:, lo[0]:lo[0] + n[0], lo[1]:lo[1] + n[1], lo[2]:lo[2] + n[2]] = r[:, :n[0], :n[1], :n[2]]
    ok = True

    def cost(valid_box):      # rounds dealt from a cost-sorted list (second repetition): ragged edge chunks last, same volume
        return int(np.prod(np.subtract(valid_box[1], valid_box[0])))
    for pipelined in (True, False):
        for root_computes in (True, False):
            for rep in range(2):                  # twice: the second call reuses the pinned pool
                trace = []
                out = par.predict_volume_distributed(vol if rank == 0 else None, vol_shape, chunk, halo, predict_fn, n_out=2,
                                                     device=dev, pipelined=pipelined, root_computes=root_computes, trace=trace,
                                                     chunk_cost=cost if rep else None)
                torch.cuda.synchronize(dev)
                nr = -(-27 // (world if root_computes else max(1, world - 1)))
                assert nr >= 5
                if rank == 0:
                    assert out is not None and torch.equal(out, want), (pipelined, root_computes, rep)
                else:
                    assert out is None
                pos = {e: i for i, e in enumerate(trace)}
                for r in range(nr):
                    assert pos[('scatter', r)] < pos[('predict', r)] < pos[('gather', r)] < pos[('stitch', r)]
                    if pipelined and r + 1 < nr:      # the next round's scatter is issued before this round's kernels
                        assert pos[('scatter', r + 1)] < pos[('predict', r)], (r, trace)
                    if not pipelined and r + 1 < nr:
                        assert pos[('stitch', r)] < pos[('scatter', r + 1)]
    par.barrier()
    if rank == 0:
        print('DIST_GPU_WORKER_OK backend=%s world=%d' % ('nccl' if two else 'gloo-staged', world))
    torch.distributed.destroy_process_group()
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
