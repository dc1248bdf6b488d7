"""Error types of the boundary, same names and bases as pufferlib/exceptions.py:5-22."""


class APIUsageError(RuntimeError):
    """Raised when the vecenv / trainer API is used incorrectly."""

    def __init__(self, message='API usage error.'):
        self.message = message
        super().__init__(self.message)


class InvalidAgentError(ValueError):
    def __init__(self, agent_id, agents):
        super().__init__(f'Invalid agent/team ({agent_id}) specified. Valid values:\n{agents}')


class ExtensionError(RuntimeError):
    """The HIP extension is missing or a kernel call failed.  Never swallowed: there is no CPU fallback."""
