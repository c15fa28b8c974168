"""`python -m covid19uk.inference.inference -c config.yaml -o posterior.hd5 data.nc`"""
from covid19uk_amd.inference.inference import main, mcmc  # noqa: F401

if __name__ == "__main__":
    main()
