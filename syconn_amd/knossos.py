"""Minimal in-repo stand-in for ``knossos_utils`` (KnossosDataset + chunky.ChunkDataset).

``knossos_utils`` (branch ``syconn2``, /root/reference/environment.yml:88) is a third-party dependency that is
neither vendored in the reference nor installed on the build / GPU boxes, so the dense path cannot do volume I/O
without this file.  It implements only what /root/reference/syconn/handler/prediction.py:666-706, 753-767, 806,
835-843 call, with the published KNOSSOS raw-cube layout (marked "restated, not verified
against source" there):

* ``<root>/mag<M>/x%04d/y%04d/z%04d/<exp>_mag<M>_x%04d_y%04d_z%04d.raw`` -- uint8 cubes, x fastest;
* ``<root>/mag<M>/knossos.conf`` with experiment name / boundary / scale / magnification;
* ``load_raw(size, offset, mag)``: `size`, `offset` in mag-1 voxels (x,y,z); returns a (z,y,x) uint8 array of
  ``size // mag`` voxels, zeros outside the dataset;
* ``save_raw / save_seg(offset, mags, data, data_mag, fast_resampling, upsample)``: `data` is (z,y,x) at `data_mag`;
  written to every mag in `mags` >= data_mag by order-0 (strided) down-sampling.

Overlay (segmentation) cubes use the KNOSSOS on-disk format: one zip archive
``<cube>.seg.sz.zip`` per cube whose single member ``<cube>.seg.sz`` is the Snappy raw-format compression of the
little-endian uint64 cube (z,y,x with x fastest).  The codec is the library's C implementation
(``include/syconn_dense.h: sd_snappy_*``; python-snappy is not installed here).  Cubes written as raw uint64
(``*.seg.raw``) by earlier versions of this file are still read.
"""
import contextlib
import fcntl
import itertools
import os
import re
import zipfile
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np


# cube files are independent: a box is read / written with one task per cube (file I/O, numpy copies and the C snappy
# codec all release the GIL)
_IO_THREADS = max(1, min(16, int(os.environ.get('SYCONN_AMD_IO_THREADS', '8'))))
_pool = None
_pool_pid = None


def _map_cubes(fn, items):
    global _pool, _pool_pid
    items = list(items)
    if _IO_THREADS == 1 or len(items) < 2:
        for it in items:
            fn(*it)
        return
    if _pool is None or _pool_pid != os.getpid():      # a forked child inherits the pool object but not its threads
        _pool = ThreadPoolExecutor(max_workers=_IO_THREADS)
        _pool_pid = os.getpid()
    for f in [_pool.submit(fn, *it) for it in items]:
        f.result()


@contextlib.contextmanager
def _cube_lock(fn: str):
    """Exclusive inter-process lock on ONE cube file for a read-modify-write.  Workers of ``predict_dense_to_kd`` own
    chunks (482x481x236 by default) that are not aligned to the 256^3 target cubes, so neighbouring chunks of DIFFERENT
    worker processes update the same cube file at about the same time; ``os.replace`` only makes the file swap atomic,
    not the read-modify-write (upstream knossos_utils serialises cube writes per cube as well).  ``flock`` on a sidecar
    file: held per open file description, so it also serialises the I/O threads of one process.

    The sidecar is REMOVED again by the holder before it releases the lock (a dataset is not littered with one lock file
    per cube); a waiter that acquires the lock on a meanwhile-unlinked file notices (inode check) and starts over.
    Requirement: the file system must implement ``flock`` across the processes / nodes that write one dataset (local file
    systems, NFSv4, Lustre mounted with ``-o flock``).  A file system that refuses it (ENOLCK / ENOTSUP / EOPNOTSUPP) is a
    hard error -- silently running without the lock would lose updates."""
    import errno
    os.makedirs(os.path.dirname(fn), exist_ok=True)
    lock = fn + '.lock'
    while True:
        fd = os.open(lock, os.O_CREAT | os.O_RDWR, 0o666)
        try:
            fcntl.flock(fd, fcntl.LOCK_EX)
        except OSError as e:
            os.close(fd)
            if e.errno in (errno.ENOLCK, errno.ENOTSUP, getattr(errno, 'EOPNOTSUPP', errno.ENOTSUP)):
                raise RuntimeError(f'{lock}: this file system does not support flock(); concurrent workers would lose cube '
                                   f'updates.  Mount it with flock support or run one worker per dataset.') from e
            raise
        try:
            same = os.fstat(fd).st_ino == os.stat(lock).st_ino
        except FileNotFoundError:
            same = False
        if same:
            break
        os.close(fd)                              # the previous holder unlinked this file: lock the new one
    try:
        yield
    finally:
        try:
            os.unlink(lock)
        except FileNotFoundError:
            pass
        try:
            fcntl.flock(fd, fcntl.LOCK_UN)
        finally:
            os.close(fd)


class KnossosDataset:
    def __init__(self):
        self._cube_shape = (128, 128, 128)
        self._experiment_name = None
        self._boundary = np.zeros(3, dtype=np.int64)
        self._scale = np.ones(3, dtype=np.float64)
        self._knossos_path = None
        self._mags: List[int] = [1]
        self.scales = []
        self._initialized = False
        self._wc = None                  # write-combining cache (enable_write_combining)

    # -- properties used by the dense path ---------------------------------------------------------
    @property
    def boundary(self) -> np.ndarray:
        return self._boundary

    @property
    def scale(self) -> np.ndarray:
        return self._scale

    @property
    def experiment_name(self) -> str:
        return self._experiment_name

    @property
    def knossos_path(self) -> str:
        return self._knossos_path

    @property
    def available_mags(self):
        return list(self._mags)

    @property
    def cube_shape(self):
        return tuple(int(c) for c in self._cube_shape)

    # -- initialisation ----------------------------------------------------------------------------
    @staticmethod
    def _parse_conf(path: str) -> Dict:
        txt = open(path).read()
        out = {'boundary': np.zeros(3, dtype=np.int64), 'scale': np.ones(3), 'mag': 1, 'name': None}
        m = re.search(r'experiment name\s+"([^"]*)"', txt)
        if m:
            out['name'] = m.group(1)
        for i, ax in enumerate('xyz'):
            m = re.search(rf'boundary {ax}\s+(\d+)', txt)
            if m:
                out['boundary'][i] = int(m.group(1))
            m = re.search(rf'scale {ax}\s+([0-9.eE+-]+)', txt)
            if m:
                out['scale'][i] = float(m.group(1))
        m = re.search(r'magnification\s+(\d+)', txt)
        if m:
            out['mag'] = int(m.group(1))
        if out['name'] is None or not np.all(out['boundary'] > 0):
            raise ValueError(f'{path}: not a knossos.conf (needs `experiment name` and a positive `boundary x/y/z`)')
        return out

    def initialize_from_knossos_path(self, path: str, **_):
        """`path`: a ``knossos.conf`` file, a ``mag*`` folder or the dataset root."""
        path = os.path.expanduser(path)
        if os.path.isfile(path):
            conf = path
        elif os.path.isfile(os.path.join(path, 'knossos.conf')):
            conf = os.path.join(path, 'knossos.conf')
        elif os.path.isfile(os.path.join(path, 'mag1', 'knossos.conf')):
            conf = os.path.join(path, 'mag1', 'knossos.conf')
        else:
            raise ValueError(f'Could not find knossos.conf at {path}.')
        c = self._parse_conf(conf)
        root = os.path.dirname(os.path.dirname(os.path.abspath(conf)))
        self._knossos_path = root + '/'
        self._experiment_name = c['name']
        self._boundary = c['boundary'] * c['mag']   # the mag-M conf stores the boundary in mag-M voxels
        self._scale = c['scale'] / c['mag']
        self._mags = sorted(int(m.group(1)) for d in os.listdir(root) for m in [re.fullmatch(r'mag(\d+)', d)] if m)
        cs = os.path.join(root, 'cube_shape.txt')
        if os.path.isfile(cs):
            self._cube_shape = tuple(int(v) for v in open(cs).read().split())
        self._initialized = True
        return self

    def initialize_from_pyknossos_path(self, path: str, **_):
        """`path`: a ``*.pyk.conf`` (the INI file pyKNOSSOS / knossos_utils write next to the ``mag*`` folders; format recalled from
        knossos_utils, un-vendored: section ``[Dataset]`` with ``_BaseName``, ``_DataScale`` = one x,y,z triple per magnification,
        ``_Extent`` = mag-1 boundary, optional ``_CubeSize`` / ``_BaseExt``).  Anything that does not carry these keys is refused --
        a silently empty dataset (boundary 0,0,0) would make every prediction a no-op."""
        import configparser
        path = os.path.expanduser(path)
        cp = configparser.ConfigParser()
        cp.optionxform = str
        try:
            with open(path) as f:
                cp.read_file(f)
        except (configparser.Error, OSError) as e:
            raise ValueError(f'{path}: not a readable pyknossos conf ({e})') from None
        if 'Dataset' not in cp:
            raise ValueError(f'{path}: no [Dataset] section -- not a pyknossos conf')
        ds = cp['Dataset']
        missing = [k for k in ('_BaseName', '_DataScale', '_Extent') if k not in ds]
        if missing:
            raise ValueError(f'{path}: pyknossos conf without {missing}')
        try:
            scales = [float(v) for v in ds['_DataScale'].replace(';', ',').split(',') if v.strip()]
            extent = [int(float(v)) for v in ds['_Extent'].split(',')]
            cube = [int(v) for v in ds.get('_CubeSize', '128,128,128').split(',')]
        except ValueError:
            raise ValueError(f'{path}: malformed _DataScale / _Extent / _CubeSize') from None
        if len(scales) < 3 or len(scales) % 3 or len(extent) != 3 or len(cube) != 3 or min(extent) <= 0:
            raise ValueError(f'{path}: _DataScale needs x,y,z per magnification and _Extent a positive x,y,z')
        if ds.get('_ServerFormat', 'knossos').strip().lower() not in ('knossos', 'pyknossos', ''):
            raise ValueError(f'{path}: server format {ds["_ServerFormat"]!r} is not a local cube store')
        root = os.path.dirname(os.path.abspath(path))
        self._knossos_path = root + '/'
        self._experiment_name = ds['_BaseName'].strip()
        self._boundary = np.asarray(extent, dtype=np.int64)
        self._scale = np.asarray(scales[:3], dtype=np.float64)
        # magnifications: the scale triples relative to the first one (isotropic powers of two in every SyConn dataset)
        self._mags = sorted({int(round(scales[3 * i] / scales[0])) for i in range(len(scales) // 3)})
        self._cube_shape = tuple(cube)
        self._initialized = True
        return self

    def initialize_from_conf(self, path: str, **_):
        """A ``knossos.conf`` or a ``*.pyk.conf`` file (by name, then by content)."""
        if str(path).endswith('.pyk.conf') or (os.path.isfile(path) and '[Dataset]' in open(path).read(4096)):
            return self.initialize_from_pyknossos_path(path)
        return self.initialize_from_knossos_path(path)

    def initialize_without_conf(self, path: str, boundary, scale, experiment_name: str, mags=None,
                                make_mag_folders: bool = True, create_knossos_conf: bool = True,
                                create_pyk_conf: bool = False, **_):
        self._knossos_path = os.path.abspath(os.path.expanduser(path)) + '/'
        self._boundary = np.asarray(boundary, dtype=np.int64)
        self._scale = np.asarray(scale, dtype=np.float64)
        self._experiment_name = experiment_name
        self._mags = [1] if mags is None else [int(m) for m in mags]
        os.makedirs(self._knossos_path, exist_ok=True)
        with open(self._knossos_path + 'cube_shape.txt', 'w') as f:
            f.write(' '.join(str(int(c)) for c in self._cube_shape))
        for mag in self._mags:
            d = f'{self._knossos_path}mag{mag}/'
            if make_mag_folders:
                os.makedirs(d, exist_ok=True)
            if create_knossos_conf:
                b = self._boundary // mag
                s = self._scale * mag
                with open(d + 'knossos.conf', 'w') as f:
                    f.write(f'experiment name "{experiment_name}";\n')
                    for i, ax in enumerate('xyz'):
                        f.write(f'boundary {ax} {int(b[i])};\n')
                    for i, ax in enumerate('xyz'):
                        f.write(f'scale {ax} {float(s[i])};\n')
                    f.write(f'magnification {mag};\n')
        if create_pyk_conf:
            with open(f'{self._knossos_path}{experiment_name}.pyk.conf', 'w') as f:
                f.write('[Dataset]\n')
                f.write(f'_BaseName = {experiment_name}\n_ServerFormat = knossos\n')
                f.write('_DataScale = ' + ', '.join(','.join(str(float(v)) for v in self._scale * mag) for mag in self._mags) + '\n')
                f.write('_Extent = ' + ','.join(str(int(v)) for v in self._boundary) + '\n')
                f.write('_CubeSize = ' + ','.join(str(int(c)) for c in self._cube_shape) + '\n')
                f.write('_BaseExt = .raw\n_FileType = 2\n_Origin = 0,0,0\n')
        self._initialized = True
        return self

    # -- cube I/O ------------------------------------------------------------------------------------
    def _cube_file(self, mag: int, cx: int, cy: int, cz: int, ext: str) -> str:
        return (f'{self._knossos_path}mag{mag}/x{cx:04d}/y{cy:04d}/z{cz:04d}/'
                f'{self._experiment_name}_mag{mag}_x{cx:04d}_y{cy:04d}_z{cz:04d}.{ext}')

    # one cube <-> one file.  ext 'raw': plain uint8; ext 'seg.sz': zip archive <file>.zip with the snappy stream
    def _read_cube(self, fn: str, ext: str, dtype, shape) -> Optional[np.ndarray]:
        if ext == 'seg.sz':
            if os.path.isfile(fn + '.zip'):
                from . import _lib
                with zipfile.ZipFile(fn + '.zip', 'r') as zf:
                    blob = zf.read(os.path.basename(fn))
                return np.frombuffer(_lib.snappy_decompress(blob), dtype=dtype).reshape(shape).copy()
            legacy = fn[:-len('seg.sz')] + 'seg.raw'
            if os.path.isfile(legacy):
                return np.fromfile(legacy, dtype=dtype).reshape(shape)
            return None
        if not os.path.isfile(fn):
            return None
        return np.fromfile(fn, dtype=dtype).reshape(shape)

    def _write_cube(self, fn: str, ext: str, cube: np.ndarray):
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        if ext == 'seg.sz':
            from . import _lib
            tmp = fn + f'.zip.tmp{os.getpid()}'
            with zipfile.ZipFile(tmp, 'w', zipfile.ZIP_DEFLATED) as zf:
                zf.writestr(os.path.basename(fn), _lib.snappy_compress(np.ascontiguousarray(cube).tobytes()))
            os.replace(tmp, fn + '.zip')
            return
        tmp = fn + f'.tmp{os.getpid()}'
        cube.tofile(tmp)
        os.replace(tmp, fn)

    def _load(self, size, offset, mag: int, ext: str, dtype, out: Optional[np.ndarray] = None) -> np.ndarray:
        size = np.asarray(size, dtype=np.int64) // mag
        off = np.asarray(offset, dtype=np.int64) // mag
        if out is None:
            out = np.zeros(tuple(size[::-1]), dtype=dtype)   # z,y,x
        else:                                             # caller-owned (e.g. page-locked) buffer, reused chunk after chunk
            if tuple(out.shape) != tuple(size[::-1]) or out.dtype != np.dtype(dtype) or not out.flags.c_contiguous:
                raise ValueError(f'load(out=...): need a C-contiguous {np.dtype(dtype)} array of shape {tuple(int(v) for v in size[::-1])}, got '
                                 f'{out.dtype} {tuple(out.shape)}')
            out.fill(0)
        cs = np.asarray(self._cube_shape, dtype=np.int64)
        bnd = self._boundary // mag
        lo = np.maximum(off, 0)
        hi = np.minimum(off + size, bnd)
        if np.any(hi <= lo):
            return out
        c_lo, c_hi = lo // cs, (hi - 1) // cs
        def one(cx, cy, cz):
            cube = self._read_cube(self._cube_file(mag, cx, cy, cz, ext), ext, dtype, tuple(cs[::-1]))
            if cube is None:
                return
            c0 = np.array([cx, cy, cz]) * cs
            a = np.maximum(lo, c0)
            b = np.minimum(hi, c0 + cs)
            src = tuple(slice(int(a[i] - c0[i]), int(b[i] - c0[i])) for i in (2, 1, 0))
            dst = tuple(slice(int(a[i] - off[i]), int(b[i] - off[i])) for i in (2, 1, 0))
            out[dst] = cube[src]                      # disjoint destination boxes

        _map_cubes(one, itertools.product(range(c_lo[0], c_hi[0] + 1), range(c_lo[1], c_hi[1] + 1),
                                          range(c_lo[2], c_hi[2] + 1)))
        return out

    def load_raw(self, size, offset, mag: int = 1, out: Optional[np.ndarray] = None, **_) -> np.ndarray:
        """`out` (extension): a C-contiguous uint8 (z,y,x) array of the result's shape to fill instead of a fresh allocation."""
        return self._load(size, offset, mag, 'raw', np.uint8, out=out)

    def load_seg(self, size, offset, mag: int = 1, **_) -> np.ndarray:
        return self._load(size, offset, mag, 'seg.sz', np.uint64)

    def _save(self, offset, mags: Sequence[int], data: np.ndarray, data_mag: int, ext: str, dtype,
              fast_resampling: bool = True, upsample: bool = False):
        off1 = np.asarray(offset, dtype=np.int64)
        cs = np.asarray(self._cube_shape, dtype=np.int64)
        for mag in mags:
            if mag < data_mag:
                if not upsample:
                    continue
                raise NotImplementedError('upsampling is not used by the dense path')
            r = mag // data_mag
            d = data if r == 1 else data[::r, ::r, ::r]      # order-0 ("fast") resampling
            # (integer data keeps its own width here -- uint8 labels for a uint64 overlay dataset are widened cube by cube in the
            # assignments below, not as one 8x larger temporary of the whole chunk)
            d = np.asarray(d) if np.asarray(d).dtype.kind in 'ui' else np.ascontiguousarray(d, dtype=dtype)
            off = off1 // mag
            size = np.asarray(d.shape[::-1], dtype=np.int64)
            bnd = self._boundary // mag
            lo, hi = np.maximum(off, 0), np.minimum(off + size, bnd)
            if np.any(hi <= lo):
                continue
            c_lo, c_hi = lo // cs, (hi - 1) // cs
            def one(cx, cy, cz, mag=mag, d=d, off=off, lo=lo, hi=hi, bnd=bnd):
                fn = self._cube_file(mag, cx, cy, cz, ext)
                c0 = np.array([cx, cy, cz]) * cs
                a, b = np.maximum(lo, c0), np.minimum(hi, c0 + cs)
                whole = np.all(a == c0) and np.all(b == c0 + cs)
                dst = tuple(slice(int(a[i] - c0[i]), int(b[i] - c0[i])) for i in (2, 1, 0))
                src = tuple(slice(int(a[i] - off[i]), int(b[i] - off[i])) for i in (2, 1, 0))
                if self._wc is not None and not whole:
                    need = int(np.prod(np.minimum(c0 + cs, bnd) - c0))      # voxels of this cube inside the dataset
                    self._wc_add((mag, ext, cx, cy, cz), fn, dtype, tuple(cs[::-1]), dst, d[src], need)
                    return
                with _cube_lock(fn):                  # read-modify-write of a cube shared with other workers
                    cube = None if whole else self._read_cube(fn, ext, dtype, tuple(cs[::-1]))
                    if cube is None:
                        cube = np.zeros(tuple(cs[::-1]), dtype=dtype)
                    cube[dst] = d[src]
                    self._write_cube(fn, ext, cube)

            _map_cubes(one, itertools.product(range(c_lo[0], c_hi[0] + 1), range(c_lo[1], c_hi[1] + 1),
                                              range(c_lo[2], c_hi[2] + 1)))

    # ---- write combining ---------------------------------------------------------------------------------------------
    # Chunks (482 x 481 x 236 by default) are not aligned to the target cubes (256^3 at three mags): written chunk by chunk
    # every cube is read, patched and rewritten about four times.  With write combining the partial cubes a process has
    # touched stay in memory; a cube whose voxels (inside the dataset) have all arrived is written ONCE without being read,
    # the rest is merged into the file (read-modify-write under the cube lock, only the boxes this process wrote) when the
    # cache overflows or at `flush()`.  Regions of different chunks never overlap, so the final dataset is the same.
    def enable_write_combining(self, max_cubes: Optional[int] = None, chunk_shape=None):
        """`max_cubes`: partially assembled cubes kept in memory (all mags together).  Default: what two chunks of `chunk_shape`
        (x,y,z at mag 1; default 482 x 481 x 236, prediction.py:674) can touch at the three mags of a save_* call -- a smaller
        cache evicts inside every chunk and most cubes take the read-modify-write path instead of the single write."""
        import collections
        import threading
        if max_cubes is None:
            cs = np.asarray(self._cube_shape, dtype=np.int64)
            ch = np.asarray((482, 481, 236) if chunk_shape is None else chunk_shape, dtype=np.int64)
            max_cubes = 2 * sum(int(np.prod(-(-(-(-ch // m)) // cs) + 1)) for m in (1, 2, 4))
        self._wc = collections.OrderedDict()
        self._wc_max = max(4, int(max_cubes))
        self._wc_mutex = threading.Lock()
        # written-out cube buffers are kept and re-zeroed for the next cube: a fresh 16 - 134 MB array per cube is a fresh mmap
        # whose first touch costs more (page faults) than the copy into it
        self._wc_free = {}

    def _wc_buffer(self, shape, dtype):
        key = (tuple(shape), np.dtype(dtype).str)
        with self._wc_mutex:
            free = self._wc_free.get(key)
            buf = free.pop() if free else None
        if buf is None:
            return np.zeros(shape, dtype=dtype)
        buf.fill(0)
        return buf

    def _wc_recycle(self, buf):
        key = (tuple(buf.shape), buf.dtype.str)
        with self._wc_mutex:
            free = self._wc_free.setdefault(key, [])
            if len(free) < 8:
                free.append(buf)

    def _wc_add(self, key, fn, dtype, shape, dst, block, need):
        import threading
        while True:
            evict = None
            with self._wc_mutex:                  # the table only: the block copy below runs under the ENTRY's lock
                ent = self._wc.get(key)
                if ent is None:
                    ent = self._wc[key] = dict(fn=fn, ext=key[1], dtype=dtype, cube=None, boxes=[], have=0, need=need,
                                               lock=threading.Lock(), dead=False)
                    if len(self._wc) > self._wc_max:
                        old = next(iter(self._wc))
                        if old != key:
                            evict = self._wc.pop(old)
                self._wc.move_to_end(key)
            if evict is not None:
                self._wc_merge(evict)
            done = None
            with ent['lock']:
                if ent['dead']:                   # evicted / completed by another thread between the lookup and here: start over
                    continue
                if ent['cube'] is None:
                    ent['cube'] = self._wc_buffer(shape, dtype)
                ent['cube'][dst] = block
                ent['boxes'].append(dst)
                ent['have'] += int(block.size)
                if ent['have'] >= ent['need']:
                    ent['dead'] = True
                    done = ent
            if done is None:
                return
            with self._wc_mutex:
                if self._wc.get(key) is done:
                    del self._wc[key]
            # "complete" only if the boxes really tile the cube: a region written twice would be counted twice
            bx = done['boxes']
            disjoint = all(any(a[i].stop <= b[i].start or b[i].stop <= a[i].start for i in range(3))
                           for n_, a in enumerate(bx) for b in bx[n_ + 1:])
            if disjoint:                          # nobody else writes into this cube: one write, no read
                with _cube_lock(done['fn']):
                    self._write_cube(done['fn'], done['ext'], done['cube'])
                self._wc_recycle(done['cube'])
                done['cube'] = None
            else:
                self._wc_merge(done, locked=True)
            return

    def _wc_merge(self, ent, locked: bool = False):
        if not locked:
            with ent['lock']:
                if ent['dead']:
                    return
                ent['dead'] = True
        if ent['cube'] is None:
            return
        with _cube_lock(ent['fn']):
            cube = self._read_cube(ent['fn'], ent['ext'], ent['dtype'], ent['cube'].shape)
            if cube is None:
                cube = ent['cube']                # (zeros outside the boxes written here)
            else:
                for box in ent['boxes']:
                    cube[box] = ent['cube'][box]
            self._write_cube(ent['fn'], ent['ext'], cube)
        self._wc_recycle(ent['cube'])
        ent['cube'] = None

    def flush(self):
        """Write out every partially assembled cube (no-op without write combining)."""
        if self._wc is None:
            return
        with self._wc_mutex:
            pending = list(self._wc.values())
            self._wc.clear()
        _map_cubes(lambda e: self._wc_merge(e), [(e,) for e in pending])

    def save_raw(self, offset, mags, data, data_mag: int = 1, fast_resampling: bool = True, upsample: bool = True,
                 **_):
        self._save(offset, mags, data, data_mag, 'raw', np.uint8, fast_resampling, upsample)

    def save_seg(self, offset, mags, data, data_mag: int = 1, fast_resampling: bool = True, upsample: bool = True,
                 **_):
        self._save(offset, mags, data, data_mag, 'seg.sz', np.uint64, fast_resampling, upsample)


class Chunk:
    def __init__(self, number: int, coordinates, size, overlap):
        self.number = number
        self.coordinates = np.asarray(coordinates, dtype=np.int64)
        self.size = np.asarray(size, dtype=np.int64)
        self.overlap = np.asarray(overlap, dtype=np.int64)


class ChunkDataset:
    """Regular chunk grid over a box (``knossos_utils.chunky.ChunkDataset.initialize`` as called at
    prediction.py:679-683 / 757-760).  Chunk ids enumerate the grid x-outermost, z-innermost."""

    def __init__(self):
        self.chunk_dict: Dict[int, Chunk] = {}
        self.box_size = None
        self.chunk_size = None
        self.box_coords = None
        self.overlap = None
        self.path_head_folder = None

    def initialize(self, knossos_dataset_object, box_size, chunk_size, path_head_folder: str, overlap=(0, 0, 0),
                   list_of_coords: Optional[Iterable] = None, box_coords=None, fit_box_size: bool = False):
        chunk_size = np.asarray(chunk_size, dtype=np.int64)
        box_size = np.asarray(box_size, dtype=np.int64)
        box_coords = np.zeros(3, dtype=np.int64) if box_coords is None else np.asarray(box_coords, dtype=np.int64)
        if fit_box_size:
            box_size = (np.ceil(box_size / chunk_size) * chunk_size).astype(np.int64)
        self.box_size, self.chunk_size, self.box_coords = box_size, chunk_size, box_coords
        self.overlap = np.asarray(overlap, dtype=np.int64)
        self.path_head_folder = path_head_folder
        self.chunk_dict = {}
        n = 0
        if list_of_coords:
            for c in list_of_coords:
                self.chunk_dict[n] = Chunk(n, c, chunk_size, self.overlap)
                n += 1
            return self
        for x in range(int(box_coords[0]), int(box_coords[0] + box_size[0]), int(chunk_size[0])):
            for y in range(int(box_coords[1]), int(box_coords[1] + box_size[1]), int(chunk_size[1])):
                for z in range(int(box_coords[2]), int(box_coords[2] + box_size[2]), int(chunk_size[2])):
                    self.chunk_dict[n] = Chunk(n, (x, y, z), chunk_size, self.overlap)
                    n += 1
        return self
