"""rat_amd — MI355X-native hot path of RAT_m2 behind the FuxiCTR model-plugin API."""
__version__ = "0.1.0"
