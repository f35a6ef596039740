"""Empty stand-in: PyTables is only touched by the reference's .h5 load/save
methods, which the golden generator never calls."""
