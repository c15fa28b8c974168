"""`python -m covid19uk.posterior.reproduction_number` -- same CLI as the reference's covid19uk/posterior/reproduction_number.py."""
from covid19uk_amd.posterior.reproduction_number import *  # noqa: F401,F403
from covid19uk_amd.posterior.reproduction_number import main  # noqa: F401

if __name__ == "__main__":
    main()
