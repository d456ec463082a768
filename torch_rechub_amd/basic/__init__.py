from . import activation, callback, features, initializers, layers, loss_func  # noqa: F401
