"""Importable alias of the package directory `gradientdomain-mitsuba_amd/` (a hyphen is not a legal
Python identifier).  Sub-modules resolve from that directory: `gradientdomain_mitsuba_amd.poisson`
is `gradientdomain-mitsuba_amd/poisson.py`."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gradientdomain-mitsuba_amd"))
