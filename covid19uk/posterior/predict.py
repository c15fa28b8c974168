"""`python -m covid19uk.posterior.predict` -- same CLI as the reference's covid19uk/posterior/predict.py."""
from covid19uk_amd.posterior.predict import *  # noqa: F401,F403
from covid19uk_amd.posterior.predict import main  # noqa: F401

if __name__ == "__main__":
    main()
