from .vis import *  # noqa: F401,F403
