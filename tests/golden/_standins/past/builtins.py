"""`from past.builtins import xrange` (reference utils.py:16)."""
xrange = range
