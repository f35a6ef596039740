# -*- coding: utf-8 -*-
"""Packaging of the hot-path drop-in: ``pip install -e .`` gives an ``alntools`` console script with the reference's
``bam2ec`` / ``bam2emase`` / ``ec2emase`` / ``emase2ec`` commands (``alntools/cli.py:43-113``, ``bin/alntools:29``).
libecb.so is built in-tree by ``python -m alntools_amd.build`` (hipcc, gfx950)."""
from setuptools import setup

setup(name="alntools_amd", version="0.2.0", packages=["alntools_amd"], package_data={"alntools_amd": ["libecb.so", "libbamdec.so", "csrc/*.hip", "csrc/*.inc", "csrc/*.c"]},
      scripts=["bin/alntools"], entry_points={"console_scripts": ["alntools-amd=alntools_amd.cli:cli"]},
      install_requires=["click", "numpy"])
