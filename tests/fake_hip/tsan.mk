# ... and under the thread sanitizer: make -C tests/fake_hip -f tsan.mk (the recipe of asan.mk with another sanitizer and tag)
SAN := -fsanitize=thread
TAG := tsan
include asan.mk
