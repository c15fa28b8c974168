"""Drop-in import name: `python -m covid19uk.inference.inference` runs the MI355X
implementation in `covid19uk_amd` (same CLI, config YAML and posterior.hd5)."""
from covid19uk_amd.inference.inference import mcmc  # noqa: F401
