# -*- coding: utf-8 -*-
"""Small helpers that fix shard contiguity and locus order (reference ``alntools/utils.py``)."""
from __future__ import annotations

import logging
from collections import OrderedDict

logging.basicConfig(format='[alntools] [%(asctime)s] %(message)s', datefmt='%m/%d/%Y %I:%M:%S %p')


def get_logger():
    """Same logger name and format as the reference (``utils.py:21-30``): users grep these lines."""
    return logging.getLogger("alntools.utils")


def configure_logging(level):
    """0 = WARNING, 1 = INFO, 2+ = DEBUG (``utils.py:33-48``)."""
    get_logger().setLevel(logging.WARN if level == 0 else logging.INFO if level == 1 else logging.DEBUG)


def format_time(start, end):
    hours, rem = divmod(end - start, 3600)
    minutes, seconds = divmod(rem, 60)
    return "{:0>2}:{:0>2}:{:05.2f}".format(int(hours), int(minutes), seconds)


def parse_targets(target_file):
    """First whitespace token of every line not starting with '#', in file order (``utils.py:161-178``)."""
    targets = OrderedDict()
    with open(target_file, 'r') as f:
        for line in f:
            if line and line[0] == '#':
                continue
            targets[line.strip().split()[0]] = len(targets)
    return targets
