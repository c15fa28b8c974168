"""`python -m covid19uk.posterior.thin` -- same CLI as the reference's covid19uk/posterior/thin.py."""
from covid19uk_amd.posterior.thin import *  # noqa: F401,F403
from covid19uk_amd.posterior.thin import main  # noqa: F401

if __name__ == "__main__":
    main()
