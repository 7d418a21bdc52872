"""Import-name shim package."""
