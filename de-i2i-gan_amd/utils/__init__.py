"""Host-side helpers of the reference's utils/ that sit on a train step."""
