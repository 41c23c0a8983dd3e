"""TEST INFRASTRUCTURE ONLY -- pure-Python restatement of the Snappy raw format (google/snappy
format_description.txt), the codec of the KNOSSOS ``*.seg.sz.zip`` overlay cubes that knossos_utils writes for
``KnossosDataset.save_seg`` (called at /root/reference/syconn/handler/prediction.py:835-843; SURVEY.md section 8f
row 1).  Upstream pins snappy 1.1.8 / python-snappy 0.6.0
(/root/reference/examples/working_env_glibc_2_27_2021_11.yml:247,270); neither is vendored in /root/reference nor
installed here, so byte-level parity of the ENCODER with upstream is UNPINNED (the format does not require it: any
conforming stream is valid).  The decoder has exactly one correct answer per stream and is what the tests pin the
C codec (syconn_amd/csrc/sd_snappy.cpp) against, together with hand-assembled streams for every element type.

Only tests/ may import this module; the product path uses the C codec and fails without the built library.
"""


def varint(n: int) -> bytes:
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7f) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def literal(data: bytes) -> bytes:
    """One literal element (format_description.txt section 2.1)."""
    n = len(data) - 1
    if n < 60:
        return bytes([n << 2]) + data
    nb = (n.bit_length() + 7) // 8
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, 'little') + data


def copy1(length: int, offset: int) -> bytes:
    """Copy with 1-byte offset (section 2.2.1): 4 <= length <= 11, 0 <= offset < 2048."""
    assert 4 <= length <= 11 and 0 <= offset < 2048
    return bytes([1 | ((length - 4) << 2) | ((offset >> 8) << 5), offset & 0xff])


def copy2(length: int, offset: int) -> bytes:
    """Copy with 2-byte offset (section 2.2.2): 1 <= length <= 64, offset < 65536."""
    assert 1 <= length <= 64 and 0 <= offset < 65536
    return bytes([2 | ((length - 1) << 2)]) + offset.to_bytes(2, 'little')


def copy4(length: int, offset: int) -> bytes:
    """Copy with 4-byte offset (section 2.2.3)."""
    assert 1 <= length <= 64 and 0 <= offset < 2 ** 32
    return bytes([3 | ((length - 1) << 2)]) + offset.to_bytes(4, 'little')


def compress_literal_only(data: bytes) -> bytes:
    """A valid (if useless) encoder: preamble + one literal per 64 KiB."""
    out = bytearray(varint(len(data)))
    for i in range(0, len(data), 65536):
        out += literal(data[i:i + 65536])
    return bytes(out)


def decompress_ref(stream: bytes) -> bytes:
    """Decoder; raises ValueError on anything the format forbids."""
    pos, shift, ulen = 0, 0, 0
    while True:
        if pos >= len(stream) or pos >= 5:
            raise ValueError('bad length preamble')
        b = stream[pos]
        pos += 1
        ulen |= (b & 0x7f) << shift
        shift += 7
        if not b & 0x80:
            break
    if ulen >= 2 ** 32:
        raise ValueError('length does not fit 32 bits')
    out = bytearray()
    n = len(stream)
    while pos < n:
        tag = stream[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            length = (tag >> 2) + 1
            if length > 60:
                nb = length - 60
                if pos + nb > n:
                    raise ValueError('truncated literal length')
                length = int.from_bytes(stream[pos:pos + nb], 'little') + 1
                pos += nb
            if pos + length > n or len(out) + length > ulen:
                raise ValueError('literal overruns input or output')
            out += stream[pos:pos + length]
            pos += length
            continue
        if kind == 1:
            if pos + 1 > n:
                raise ValueError('truncated copy')
            length = 4 + ((tag >> 2) & 7)
            offset = ((tag >> 5) << 8) | stream[pos]
            pos += 1
        elif kind == 2:
            if pos + 2 > n:
                raise ValueError('truncated copy')
            length = (tag >> 2) + 1
            offset = int.from_bytes(stream[pos:pos + 2], 'little')
            pos += 2
        else:
            if pos + 4 > n:
                raise ValueError('truncated copy')
            length = (tag >> 2) + 1
            offset = int.from_bytes(stream[pos:pos + 4], 'little')
            pos += 4
        if offset == 0 or offset > len(out) or len(out) + length > ulen:
            raise ValueError('bad copy')
        for _ in range(length):          # byte by byte: a copy may overlap its own output
            out.append(out[-offset])
    if len(out) != ulen:
        raise ValueError('stream ends before the announced length')
    return bytes(out)
