"""Host-side helpers of the hybrid head (torch): box coder, losses, box utilities, heat-map targets."""
