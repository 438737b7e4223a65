from .factory import get_network, list_networks  # noqa: F401
