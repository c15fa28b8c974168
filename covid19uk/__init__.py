"""Drop-in import name for the hot path of chrism0dwk/covid19uk: the stage functions of the reference's
`covid19uk/__init__.py:3-21` that sit on or next to the posterior sampler resolve to the MI355X
implementation in `covid19uk_amd` (same call signatures, config YAML and file formats).

`assemble_data` (network ETL, covid19uk/data/*) is outside the scope of this build
and is not provided."""
from covid19uk_amd.inference.inference import mcmc  # noqa: F401
from covid19uk_amd.posterior.predict import predict  # noqa: F401
from covid19uk_amd.posterior.reproduction_number import reproduction_number  # noqa: F401
from covid19uk_amd.posterior.thin import thin_posterior  # noqa: F401
from covid19uk_amd.posterior.within_between import within_between  # noqa: F401

__all__ = ["mcmc", "thin_posterior", "reproduction_number", "predict", "within_between"]
