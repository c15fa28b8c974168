from covid19uk_amd.inference.inference import mcmc  # noqa: F401
