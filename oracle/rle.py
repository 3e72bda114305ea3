"""TEST-ONLY restatement of COCO run-length mask encoding.

The reference hands masks to the evaluator as `pycocotools.mask.encode(np.asfortranarray(mask))`
(mmdet/structures/mask/utils.py:38-53, consumed by CocoMetric.process, coco_metric.py:346-391).  pycocotools is a
third-party dependency absent from /root/reference (requirements/runtime.txt: `pycocotools`, unpinned); its published
algorithm (cocoapi common/maskApi.c: rleEncode, rleToString, rleFrString, rleDecode) is restated here.  Pinned by the
compressed RLE strings the reference ships in tests/data/vis_sample.json (tests/golden/coco_rle_strings.json):
decode -> encode must reproduce every string byte for byte (tests/test_oracle_golden.py).
"""
import numpy as np


def rle_counts(mask):
    """maskApi.c rleEncode: run lengths of the column-major (Fortran-order) pixel stream, starting with a zeros run."""
    t = np.asarray(mask, dtype=np.uint8).reshape(mask.shape[0], mask.shape[1]).T.reshape(-1)   # j = x*h + y
    if t.size == 0:
        return np.zeros((0,), dtype=np.uint32)
    prev = np.concatenate([[0], t[:-1]])
    pos = np.flatnonzero(t != prev)
    edges = np.concatenate([[0], pos, [t.size]])
    cnts = np.diff(edges)
    return cnts.astype(np.uint32)            # cnts[0] = leading zeros (0 when the stream starts with a 1)


def rle_to_string(cnts):
    """maskApi.c rleToString: LEB128-like, 5 data bits + continuation bit per char, offset 48; counts from the third
    on are stored as differences to the count two places earlier."""
    out = bytearray()
    cnts = [int(c) for c in cnts]
    for i, x in enumerate(cnts):
        if i > 2:
            x -= cnts[i - 2]
        more = True
        while more:
            c = x & 0x1f
            x >>= 5                           # arithmetic shift (Python ints): matches the C `long` behaviour
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return bytes(out)


def rle_from_string(s):
    """maskApi.c rleFrString."""
    if isinstance(s, str):
        s = s.encode('ascii')
    cnts = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return np.asarray(cnts, dtype=np.int64)


def rle_decode(cnts, h, w):
    """maskApi.c rleDecode -> bool [h, w]."""
    vals = np.zeros(len(cnts), dtype=np.uint8)
    vals[1::2] = 1
    t = np.repeat(vals, np.asarray(cnts, dtype=np.int64))
    assert t.size == h * w, (t.size, h, w)
    return t.reshape(w, h).T.astype(bool)


def encode(mask):
    """pycocotools.mask.encode for one [h, w] mask: dict(size=[h, w], counts=bytes)."""
    h, w = mask.shape
    return dict(size=[int(h), int(w)], counts=rle_to_string(rle_counts(mask)))
