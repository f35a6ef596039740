# -*- coding: utf-8 -*-
"""Same driver functions as the reference's ``alntools/methods.py:32-53, 205-210`` for the hot path."""
from __future__ import annotations

from . import bam_utils, bin_utils


def bam2ec(bam_filename, ec_filename, chunks=0, directory=None, number_processes=-1, range_filename=None,
           sample=None, target_filename=None):
    return bam_utils.convert(bam_filename, ec_filename, None, num_chunks=chunks, number_processes=number_processes,
                             temp_dir=directory, range_filename=range_filename, sample=sample,
                             target_filename=target_filename)


def bam2emase(bam_filename, emase_filename, chunks=0, directory=None, number_processes=-1, range_filename=None,
              target_filename=None):
    return bam_utils.convert(bam_filename, None, emase_filename, num_chunks=chunks, number_processes=number_processes,
                             temp_dir=directory, range_filename=range_filename, target_filename=target_filename)


def bam2both(bam_filename, ec_filename, emase_filename, chunks=0, directory=None, number_processes=-1,
             range_filename=None, sample=None, target_filename=None):
    return bam_utils.convert(bam_filename, ec_filename, emase_filename, num_chunks=chunks,
                             number_processes=number_processes, temp_dir=directory, range_filename=range_filename,
                             sample=sample, target_filename=target_filename)


def bam2ec_multisample(bam_filename, ec_filename, chunks=0, minimum_count=-1, directory=None, number_processes=-1,
                       range_filename=None, target_filename=None):
    from . import bam_utils_multisample
    return bam_utils_multisample.convert(bam_filename, ec_filename, None, chunks, minimum_count, number_processes,
                                         directory, range_filename, target_filename)


def bam2emase_multisample(bam_filename, emase_filename, chunks=0, minimum_count=-1, directory=None,
                          number_processes=-1, range_filename=None, target_filename=None):
    from . import bam_utils_multisample
    return bam_utils_multisample.convert(bam_filename, None, emase_filename, chunks, minimum_count, number_processes,
                                         directory, range_filename, target_filename)


def ec2emase(ec_file, emase_file):
    bin_utils.ec2emase(ec_file, emase_file)


def emase2ec(emase_file, ec_file):
    bin_utils.emase2ec(emase_file, ec_file)
