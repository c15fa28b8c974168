"""Inference stage: `mcmc(data_file, output_file, config)` and its CLI."""
from .inference import mcmc  # noqa: F401
