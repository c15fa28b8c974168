"""MI355X-native spatial SEIR posterior sampler (hot path of chrism0dwk/covid19uk)."""
__version__ = "0.1.0"
