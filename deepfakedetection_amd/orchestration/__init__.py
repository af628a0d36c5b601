"""Host-side mirror of the reference's plug-in surface (orchestration/*)."""
