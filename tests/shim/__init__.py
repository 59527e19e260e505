"""TEST INFRASTRUCTURE: stand-ins that let the REFERENCE's own classes run in this image (tests and golden generators only)."""
