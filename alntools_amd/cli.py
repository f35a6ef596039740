# -*- coding: utf-8 -*-
"""Command line with the reference's hot-path sub-commands and options (``alntools/cli.py:43-113``):
``bam2ec``, ``bam2emase``, ``ec2emase``, ``emase2ec``.  ``python -m alntools_amd.cli bam2ec in.bam out.bin``."""
from __future__ import annotations

import os

import click

# The one-GPU commands move everything through libecb's host-pointer entry points: no tensors, so PyTorch is not imported and libecb
# runs on the system's HIP runtime (alntools_amd/ecb.py: load).  ALNTOOLS_GPUS=N (> 1) runs ranks over torch.distributed and wants it.
if int(os.environ.get("ALNTOOLS_GPUS", "1") or 1) <= 1:
    os.environ.setdefault("ALNTOOLS_TORCH", "0")

from . import methods, utils  # noqa: E402


@click.group()
def cli():
    """alntools hot path on MI355X (libecb)."""


def _common(f):
    for opt in (click.option('-v', '--verbose', count=True, help='enables verbose mode'),
                click.option('-t', '--targets', default=None, type=click.Path(exists=True, dir_okay=False), help='target file'),
                click.option('--rangefile', default=None, type=click.Path(dir_okay=False), help='range file'),
                click.option('-p', '--number-processes', '--processes', 'processes', default=-1,
                             help='number of processes (accepted for compatibility: the work runs on the GPU)'),
                click.option('--multisample', is_flag=True, help='BAM_FILE is a directory of per-sample BAM files'),
                click.option('-m', '--mincount', default=None, type=int, help='minimum reads per cell (multisample)'),
                click.option('-d', '--directory', default=None, type=click.Path(file_okay=False), help='accepted for compatibility; ignored'),
                click.option('-c', '--chunks', default=0, help='accepted for compatibility; ignored')):
        f = opt(f)
    return f


@cli.command('bam2ec', short_help='convert a BAM file to EC')
@click.argument('bam_file', metavar='bam_file', type=click.Path(exists=True, resolve_path=True))
@click.argument('ec_file', metavar='ec_file', type=click.Path(resolve_path=True, dir_okay=False))
@click.option('-s', '--sample', default=None, help='sample identifier')
@_common
def bam2ec(bam_file, ec_file, chunks, directory, mincount, multisample, processes, rangefile, targets, verbose, sample):
    """Convert a BAM file (bam_file) to a binary file (ec_file)."""
    utils.configure_logging(verbose)
    if multisample:
        if sample:                                                   # cli.py:60-63
            print('-s, --sample should NOT be specified with --multisample')
            return
        methods.bam2ec_multisample(bam_file, ec_file, chunks, 1000 if mincount is None else mincount, directory,
                                   processes, rangefile, targets)
    else:
        methods.bam2ec(bam_file, ec_file, chunks, directory, processes, rangefile, sample, targets)


@cli.command('bam2emase', short_help='convert a BAM file to EMASE')
@click.argument('bam_file', metavar='bam_file', type=click.Path(exists=True, resolve_path=True))
@click.argument('emase_file', metavar='emase_file', type=click.Path(resolve_path=True, dir_okay=False))
@_common
def bam2emase(bam_file, emase_file, chunks, directory, mincount, multisample, processes, rangefile, targets, verbose):
    """Convert a BAM file (bam_file) to an EMASE file (emase_file)."""
    utils.configure_logging(verbose)
    if multisample:
        methods.bam2emase_multisample(bam_file, emase_file, chunks, 2000 if mincount is None else mincount, directory,
                                      processes, rangefile, targets)
    else:
        methods.bam2emase(bam_file, emase_file, chunks, directory, processes, rangefile, targets)


@cli.command('ec2emase', short_help='convert an EC file to EMASE')
@click.argument('ec_file', metavar='ec_file', type=click.Path(exists=True, resolve_path=True, dir_okay=False))
@click.argument('emase_file', metavar='emase_file', type=click.Path(resolve_path=True, dir_okay=False))
@click.option('-v', '--verbose', count=True, help='enables verbose mode')
def ec2emase(ec_file, emase_file, verbose):
    utils.configure_logging(verbose)
    methods.ec2emase(ec_file, emase_file)


@cli.command('emase2ec', short_help='convert an EMASE file to EC')
@click.argument('emase_file', metavar='emase_file', type=click.Path(exists=True, resolve_path=True, dir_okay=False))
@click.argument('ec_file', metavar='ec_file', type=click.Path(resolve_path=True, dir_okay=False))
@click.option('-v', '--verbose', count=True, help='enables verbose mode')
def emase2ec(emase_file, ec_file, verbose):
    utils.configure_logging(verbose)
    methods.emase2ec(emase_file, ec_file)


if __name__ == '__main__':
    cli()
