"""genjax_amd — MI355X-native inference-kernel backend with the GenJAX API shape."""
