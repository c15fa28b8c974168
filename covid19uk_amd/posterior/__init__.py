"""Posterior post-processing mirrors: thin, reproduction_number."""
