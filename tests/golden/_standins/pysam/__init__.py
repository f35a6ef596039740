"""pysam.AlignmentFile look-alike over alntools_amd.bamio.BamReader, exposing
exactly the attributes the reference reads (bam_utils.py:96-97,253-301,561-633,
754,809; bam_utils_multisample.py:209-292,399-465,657,723)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", "..")))
from alntools_amd.bamio import BamReader  # noqa: E402


class AlignedSegment(object):
    __slots__ = ("query_name", "flag", "reference_id", "reference_start", "next_reference_id",
                 "next_reference_start", "_refs")

    def __init__(self, rec, refs):
        (self.query_name, self.flag, self.reference_id, self.reference_start,
         self.next_reference_id, self.next_reference_start) = rec
        self._refs = refs

    is_paired = property(lambda s: bool(s.flag & 0x1))
    is_proper_pair = property(lambda s: bool(s.flag & 0x2))
    is_unmapped = property(lambda s: bool(s.flag & 0x4))
    is_read2 = property(lambda s: bool(s.flag & 0x80))

    @property
    def reference_name(self):
        return self._refs[self.reference_id] if self.reference_id >= 0 else None


class AlignmentFile(object):
    def __init__(self, filename, mode="rb"):
        self._r = BamReader(filename)
        self.references = self._r.references
        self.lengths = self._r.lengths
        self._tid = {}
        for i, n in enumerate(self.references):
            self._tid.setdefault(n, i)

    def get_tid(self, name):
        return self._tid.get(name, -1)

    gettid = get_tid

    def tell(self):
        # htslib: once a block is fully consumed the virtual offset is (next block, 0)
        r = self._r
        if len(r._buf) - r._pos == 0:
            return r._fh.tell() << 16
        raise NotImplementedError("tell() inside a block is not needed for single-chunk runs")

    def __next__(self):
        return AlignedSegment(self._r.__next__(), self.references)

    def __iter__(self):
        return self

    def close(self):
        self._r.close()
