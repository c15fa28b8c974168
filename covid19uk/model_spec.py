"""`covid19uk.model_spec`: constants and host-side model helpers (covid19uk/model_spec.py:22-26,108-126)."""
from covid19uk_amd.model_spec import *  # noqa: F401,F403
