"""CPU restatement of the reference's header tokeniser and per-field delta coder.

TEST INFRASTRUCTURE ONLY (tests/ may import it; the product -- fqcomp28_amd/csrc/headers.hpp --
never does).  Plain Python loops: headers are a few dozen bytes per read and the test files are
small.  Follows, function by function:
  format_from_header   HeaderFormatSpeciciation::fromHeader     src/headers.cpp:44-74
  split_header         fromHeader / encodeHeader field search   src/headers.cpp:26-42, src/workspace.cpp:95-126
  FieldStreams.store_* FieldStorageDst::storeString/Numeric     src/headers.cpp:76-91, 110-120
  FieldStreams.load_*  FieldStorageSrc::loadNextString/Numeric  src/headers.cpp:93-108, 122-133
  encode_headers       CompressionWorkspace::encodeChunk's header part (startNewChunk: the previous
                       fields of a chunk's first header are the dataset's first header)
                       src/workspace.cpp:14-31, 90-93
Pinned by the reference's own known answers (test/headers_test.cpp:12-33: the two example headers)
and its round-trip properties; the reference asserts no stream bytes anywhere.
"""
import re
import struct

NUMERIC, STRING = "N", "S"
_FROM_CHARS = re.compile(rb"-?[0-9]+")


def parse_numeric(val: bytes) -> int:
    """std::from_chars(first, last, int32) as the reference uses it (src/headers.cpp:19, 116): an optional '-',
    digits up to the first other character; no digit or outside int32 = the reference's assert (ValueError here)"""
    m = _FROM_CHARS.match(val)
    if not m:
        raise ValueError("not a number: %r" % val)
    v = int(m.group())
    if not -2**31 <= v < 2**31:
        raise ValueError("outside int32: %r" % val)
    return v


def _isalnum(c):  # std::isalnum in the "C" locale
    return (48 <= c <= 57) or (65 <= c <= 90) or (97 <= c <= 122)


def format_from_header(header: bytes):
    """-> (types, separators); raises ValueError like the reference throws invalid_argument"""
    assert header[:1] == b"@"
    types, seps = [], []
    pos = 1
    while True:
        end = pos
        while end < len(header) and _isalnum(header[end]):
            end += 1
        field = header[pos:end]
        types.append(NUMERIC if all(48 <= c <= 57 for c in field) else STRING)
        if end == len(header):
            break
        if end == len(header) - 1:
            raise ValueError("header should end in alnum char")
        seps.append(header[end])
        pos = end + 1
    return types, seps


def split_header(header: bytes, seps):
    """fields of a header: field i ends at the first separator i found from the field's SECOND byte on"""
    out, pos = [], 1
    for sep in seps:
        end = header.find(bytes([sep]), pos + 1)
        if end < 0:
            end = len(header)
        out.append(header[pos:end])
        pos = end + 1
    out.append(header[pos:])
    return out


class FieldStreams:
    def __init__(self):
        self.flags, self.content, self.lengths = bytearray(), bytearray(), bytearray()

    def store_string(self, val: bytes, prev: bytes) -> bytes:
        if val == prev:
            self.flags.append(0)
            return prev
        self.flags.append(1)
        if len(val) >= 255:  # FIELDLEN_MAX: the reference's assert (src/headers.cpp:83)
            raise ValueError("field of 255 or more bytes")
        self.content += val
        self.lengths.append(len(val))
        return val

    def store_numeric(self, val: bytes, prev: int) -> int:
        v = parse_numeric(val)
        self.content += struct.pack("<I", (v - prev) & 0xFFFFFFFF)
        return v


def encode_headers(headers, first_header=None):
    """-> (types, seps, [FieldStreams per field]) for the headers of one chunk"""
    first_header = headers[0] if first_header is None else first_header
    types, seps = format_from_header(first_header)
    prev = [parse_numeric(f) if t == NUMERIC else f for f, t in zip(split_header(first_header, seps), types)]
    streams = [FieldStreams() for _ in types]
    for h in headers:
        for i, (f, t) in enumerate(zip(split_header(h, seps), types)):
            prev[i] = streams[i].store_numeric(f, prev[i]) if t == NUMERIC else streams[i].store_string(f, prev[i])
    return types, seps, streams


def decode_headers(n, first_header, streams):
    types, seps = format_from_header(first_header)
    prev = [int(f) if t == NUMERIC else f for f, t in zip(split_header(first_header, seps), types)]
    cur = [[0, 0, 0] for _ in types]  # flag, content, length cursors
    out = []
    for _ in range(n):
        h = bytearray(b"@")
        for i, t in enumerate(types):
            s, c = streams[i], cur[i]
            if t == NUMERIC:
                (d,) = struct.unpack_from("<I", s.content, c[1]); c[1] += 4
                v = (prev[i] + d) & 0xFFFFFFFF
                prev[i] = v - (1 << 32) if v >= 1 << 31 else v
                h += str(prev[i]).encode()
            else:
                flag = s.flags[c[0]]; c[0] += 1
                if flag:
                    ln = s.lengths[c[2]]; c[2] += 1
                    prev[i] = bytes(s.content[c[1]: c[1] + ln]); c[1] += ln
                h += prev[i]
            if i + 1 < len(types):
                h.append(seps[i])
        out.append(bytes(h))
    return out
