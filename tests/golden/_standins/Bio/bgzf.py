"""Names the reference's bam_utils imports from Bio.bgzf.  Only the constants
and the virtual-offset helpers are functional; the reader/writer classes are
reached only on the multi-chunk path, which the golden generator does not use
(it is dead on Python >= 3.7 in the reference: bam_utils.py:1320-1343, PEP 479)."""
_bgzf_magic = b"\x1f\x8b\x08\x04"
_bytes_BC = b"BC"


def make_virtual_offset(block_start_offset, within_block_offset):
    return (block_start_offset << 16) | within_block_offset


def split_virtual_offset(virtual_offset):
    return virtual_offset >> 16, virtual_offset & 0xFFFF


class BgzfReader(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("multi-chunk path not exercised by the golden generator")


class BgzfWriter(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("multi-chunk path not exercised by the golden generator")
