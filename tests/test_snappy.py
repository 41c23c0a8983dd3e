"""CPU: the C Snappy codec of the library (KNOSSOS *.seg.sz.zip overlay cubes) against the
pure-Python restatement of the published format in oracle/snappy_ref.py, hand-assembled streams for every element
type, round trips over the sizes and contents label cubes produce, and the error behaviour on corrupt input."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import snappy_ref as R  # noqa: E402
from syconn_amd import _lib as L  # noqa: E402


def _cases():
    rng = np.random.default_rng(0)
    yield b''
    yield b'a'
    yield b'abc' * 5
    yield bytes(7)
    yield bytes(rng.integers(0, 256, 59, dtype=np.uint8))        # literal length boundaries 60 / 61
    yield bytes(rng.integers(0, 256, 60, dtype=np.uint8))
    yield bytes(rng.integers(0, 256, 61, dtype=np.uint8))
    yield bytes(rng.integers(0, 256, 257, dtype=np.uint8))
    yield bytes(rng.integers(0, 4, 70000, dtype=np.uint8))       # compressible noise across a 64 KiB block edge
    for n in (65535, 65536, 65537, 131072 + 5):
        yield bytes(rng.integers(0, 256, n, dtype=np.uint8))     # incompressible, block boundaries
        yield bytes(n)                                           # one long run
    lab = np.zeros((24, 40, 40), np.uint64)                      # label-volume like: few ids, large constant regions
    lab[3:9, 5:30, 7:33] = 4611686018427387905
    lab[10:20, :, 20:] = 77
    lab[rng.integers(0, 24, 50), rng.integers(0, 40, 50), rng.integers(0, 40, 50)] = rng.integers(1, 2 ** 40, 50)
    yield lab.tobytes()
    yield np.arange(30000, dtype=np.uint64).tobytes()            # every 8-byte word differs in its low bytes


@pytest.mark.parametrize('i', range(19))
def test_roundtrip_and_cross_decode(i):
    data = list(_cases())[i]
    comp = L.snappy_compress(data)
    assert len(comp) <= 32 + len(data) + len(data) // 6
    assert R.decompress_ref(comp) == data          # the oracle decodes what the C encoder wrote
    assert L.snappy_decompress(comp) == data       # the C decoder too
    assert L.snappy_decompress(R.compress_literal_only(data)) == data   # and what another valid encoder wrote


def test_label_cubes_compress():
    lab = np.zeros((128, 128, 128), np.uint64)
    lab[20:90, 30:100, 10:120] = 123456789012
    comp = L.snappy_compress(lab.tobytes())
    assert len(comp) < lab.nbytes // 15             # 64-byte copies at offset 8: ~3 bytes per 64
    assert np.array_equal(np.frombuffer(L.snappy_decompress(comp), np.uint64).reshape(lab.shape), lab)


def test_every_element_type_known_answers():
    """Streams assembled by hand from the format description -> the one correct decoding."""
    # literal + 1-byte-offset copy (length 4..11, offset < 2048)
    s = R.varint(10) + R.literal(b'abcdef') + R.copy1(4, 6)
    assert s == bytes([10, 5 << 2]) + b'abcdef' + bytes([0b00000001, 6])
    assert L.snappy_decompress(s) == b'abcdefabcd' == R.decompress_ref(s)
    # overlapping copy = run-length: "ab" repeated
    s = R.varint(12) + R.literal(b'ab') + R.copy2(10, 2)
    assert L.snappy_decompress(s) == b'ab' * 6 == R.decompress_ref(s)
    # 2-byte offset reaching far back, 4-byte offset, offset with high bits in the 1-byte form
    blob = bytes(range(256)) * 12                  # 3072 bytes
    s = R.varint(len(blob) + 64 + 8 + 11) + R.literal(blob) + R.copy2(64, 3000) + R.copy4(8, 3072) + R.copy1(11, 2047)
    want = bytearray(blob)
    for length, off in ((64, 3000), (8, 3072), (11, 2047)):
        for _ in range(length):
            want.append(want[-off])
    assert L.snappy_decompress(s) == bytes(want) == R.decompress_ref(s)
    # literal lengths with 1, 2 and 3 extra length bytes
    for n in (61, 256, 257, 65536, 70000):
        payload = bytes((i * 7) & 0xff for i in range(n))
        s = R.varint(n) + R.literal(payload)
        assert s[len(R.varint(n))] >> 2 == 59 + ((n - 1).bit_length() + 7) // 8
        assert L.snappy_decompress(s) == payload
    assert L.snappy_compress(b'') == b'\x00' and L.snappy_decompress(b'\x00') == b''


@pytest.mark.parametrize('bad', [
    b'',                                            # no preamble
    b'\xff\xff\xff\xff\xff\x7f',                    # preamble longer than 5 bytes
    R.varint(5) + R.literal(b'abc'),                # stream ends before the announced length
    R.varint(2) + R.literal(b'abc'),                # literal longer than the announced length
    R.varint(8) + R.literal(b'abcd') + bytes([0b00000001]),            # truncated copy
    R.varint(8) + R.literal(b'abcd') + R.copy1(4, 0),                  # offset 0
    R.varint(8) + R.literal(b'abcd') + R.copy2(4, 5),                  # offset before the start of the output
    R.varint(6) + R.literal(b'abcd') + R.copy1(4, 4),                  # copy overruns the announced length
    R.varint(70) + bytes([61 << 2, 0x45]),                             # literal length bytes cut off
])
def test_corrupt_streams_raise(bad):
    with pytest.raises(ValueError):
        L.snappy_decompress(bad)
    with pytest.raises(ValueError):
        R.decompress_ref(bad)
