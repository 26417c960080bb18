"""Host-side mirror of the reference's ``app`` package, limited to the neighbourhood stage."""
