"""grl/rl_modules: only the value networks the QGPO critic phase needs."""
