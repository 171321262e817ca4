"""Training strategies with the reference's import surface (train.py:43-57)."""
