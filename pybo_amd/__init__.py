"""pybo_amd -- MI355X-native GP-posterior + acquisition engine behind pybo's plugin API."""
__version__ = '0.1.0'
