__version__ = "0.5.3+mi355x.1"
