"""The legs bench.py runs after its timed region (each fenced; never part of `value`) and what they share."""
