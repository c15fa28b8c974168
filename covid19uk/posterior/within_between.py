"""`python -m covid19uk.posterior.within_between` -- same CLI as the reference's covid19uk/posterior/within_between.py."""
from covid19uk_amd.posterior.within_between import *  # noqa: F401,F403
from covid19uk_amd.posterior.within_between import main  # noqa: F401

if __name__ == "__main__":
    main()
