"""`python -m covid19uk.posterior.{thin,predict,reproduction_number,within_between}`: the reference's
post-processing entry points (covid19uk/posterior/*.py), served by `covid19uk_amd.posterior`."""
